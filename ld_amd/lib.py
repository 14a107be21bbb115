"""ctypes binding of libldhip.so (C ABI declared in include/ld_hip.h).

There is no fallback: if the shared library is missing, or a tensor handed to
an op is not a contiguous fp32/int64 tensor on a HIP device, the call raises.
"""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, '_lib', 'libldhip.so')

LD_MAX_LEVELS = 8
LD_NUM_LOSS_KEYS = 8
LOSS_KEYS = ('loss_cls', 'loss_bbox', 'loss_dfl', 'loss_ld', 'loss_ld_vlr',
             'loss_kd', 'loss_kd_neg', 'loss_im')


class LdError(RuntimeError):
    pass


class LevelT(C.Structure):
    _fields_ = [('H', C.c_int32), ('W', C.c_int32), ('stride', C.c_int32),
                ('offset', C.c_int32)]


class GeomT(C.Structure):
    _fields_ = [('num_levels', C.c_int32), ('num_anchors', C.c_int32),
                ('num_imgs', C.c_int32), ('anchor_scale', C.c_int32),
                ('lv', LevelT * LD_MAX_LEVELS)]


class MapsT(C.Structure):
    _fields_ = [('ptr', C.c_void_p * LD_MAX_LEVELS),
                ('stride_n', C.c_int64 * LD_MAX_LEVELS),
                ('stride_c', C.c_int64 * LD_MAX_LEVELS)]


class LossHpT(C.Structure):
    _fields_ = [('num_classes', C.c_int32), ('reg_max', C.c_int32),
                ('topk', C.c_int32), ('feat_channels', C.c_int32),
                ('lw_cls', C.c_float), ('qfl_beta', C.c_float),
                ('lw_bbox', C.c_float), ('giou_eps', C.c_float),
                ('lw_dfl', C.c_float), ('lw_ld', C.c_float),
                ('T_ld', C.c_float), ('lw_ld_vlr', C.c_float),
                ('T_ld_vlr', C.c_float), ('lw_kd', C.c_float),
                ('T_kd', C.c_float), ('lw_im', C.c_float),
                ('cls_channels', C.c_int32), ('flags', C.c_int32),
                ('lw_ctr', C.c_float), ('focal_alpha', C.c_float)]


LD_LOSS_PROB_CLS = 1
LD_IM_CENTER_INSIDE = 2
LD_LOSS_ATSS = 4
LD_LOSS_FCOS = 8
LD_LOSS_RETINA = 16
LD_INFER_VOTING = 1
LD_INFER_PROB = 2
LD_INFER_POINTS = 4


class ConvLevelT(C.Structure):
    _fields_ = [('Hin', C.c_int32), ('Win', C.c_int32), ('Hout', C.c_int32),
                ('Wout', C.c_int32), ('off_in', C.c_int32),
                ('off_out', C.c_int32)]


class ConvT(C.Structure):
    _fields_ = [('N', C.c_int32), ('Cin', C.c_int32), ('Cout', C.c_int32),
                ('KH', C.c_int32), ('KW', C.c_int32), ('stride', C.c_int32),
                ('pad', C.c_int32), ('num_levels', C.c_int32),
                ('Pin', C.c_int32), ('Pout', C.c_int32),
                ('lv', ConvLevelT * LD_MAX_LEVELS)]


class WtJobT(C.Structure):  # ld_wt_job_t
    _fields_ = [('w', C.c_void_p), ('wt_fwd', C.c_void_p),
                ('wt_bwd', C.c_void_p), ('Cout', C.c_int32),
                ('Cin', C.c_int32), ('ntaps', C.c_int32),
                ('first_block', C.c_int32)]


class WgradJobT(C.Structure):  # ld_wgrad_job_t
    _fields_ = [('slabs', C.c_void_p), ('dw', C.c_void_p),
                ('splits', C.c_int32), ('ntaps', C.c_int32),
                ('Cout', C.c_int32), ('Cin', C.c_int32),
                ('accumulate', C.c_int32), ('first_block', C.c_int32)]


class BnFinJobT(C.Structure):  # ld_bn_fin_job_t
    _fields_ = [('partial', C.c_void_p), ('dgamma', C.c_void_p),
                ('dbeta', C.c_void_p), ('C', C.c_int32), ('nsplit', C.c_int32),
                ('accumulate', C.c_int32), ('first_block', C.c_int32)]


LD_GRAD_DEFER = 2


class BottleneckT(C.Structure):  # ld_bottleneck_t
    _fields_ = [('N', C.c_int32), ('H', C.c_int32), ('W', C.c_int32),
                ('Cin', C.c_int32), ('mid', C.c_int32), ('reserved', C.c_int32),
                ('w1', C.c_void_p), ('w2', C.c_void_p), ('w3', C.c_void_p),
                ('scale1', C.c_void_p), ('shift1', C.c_void_p),
                ('scale2', C.c_void_p), ('shift2', C.c_void_p),
                ('scale3', C.c_void_p), ('shift3', C.c_void_p)]


class BnJobT(C.Structure):  # ld_bn_job_t
    _fields_ = [('gamma', C.c_void_p), ('beta', C.c_void_p),
                ('mean', C.c_void_p), ('var', C.c_void_p),
                ('scale', C.c_void_p), ('shift', C.c_void_p),
                ('rstd', C.c_void_p), ('eps', C.c_float), ('C', C.c_int32),
                ('first_block', C.c_int32), ('reserved', C.c_int32)]


class ImageT(C.Structure):  # ld_image_t
    _fields_ = [('data', C.c_void_p), ('src_h', C.c_int32),
                ('src_w', C.c_int32), ('new_h', C.c_int32),
                ('new_w', C.c_int32), ('flip', C.c_int32),
                ('reserved', C.c_int32)]


class ConvEpilogueT(C.Structure):
    _fields_ = [('bias', C.c_void_p), ('scale', C.c_void_p),
                ('shift', C.c_void_p), ('residual', C.c_void_p),
                ('relu', C.c_int32), ('reserved', C.c_int32),
                ('y_c8', C.c_void_p), ('residual_c8', C.c_void_p),
                ('y_raw', C.c_void_p), ('y_raw_c8', C.c_void_p)]


class LevelsT(C.Structure):
    _fields_ = [('num_levels', C.c_int32), ('H', C.c_int32 * LD_MAX_LEVELS),
                ('W', C.c_int32 * LD_MAX_LEVELS)]


_lib = None


def lib_available():
    return os.path.exists(LIB_PATH)


def get_lib():
    """Load libldhip.so (once).  Raises LdError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LdError(
            f'{LIB_PATH} not found: the HIP extension has not been built '
            '(run `python -m ld_amd.build`); ld_amd has no fallback path')
    lib = C.CDLL(LIB_PATH)
    _declare(lib)
    if lib.ld_abi_version() != ABI_VERSION:
        raise LdError('libldhip.so ABI version mismatch; rebuild')
    _load_tune_tables(lib)
    _lib = lib
    return lib


TUNE_TABLE = os.path.join(_HERE, 'tune', 'gfx950.txt')


def _load_tune_tables(lib):
    """Conv tile shapes: the table shipped in-tree (picked on an MI355X), then
    LD_CONV_TUNE_FILE on top of it.  With the same tables every process and
    every rank launches the same shapes, so results are bit-reproducible; a
    geometry in neither falls back to a deterministic model inside the
    library.  Timing new geometries is explicit: ld_amd.layers.autotune()."""
    if os.environ.get('LD_CONV_TUNE_TABLE', '1') != '0' and \
            os.path.exists(TUNE_TABLE):
        lib.ld_conv_tune_load(TUNE_TABLE.encode())
    extra = os.environ.get('LD_CONV_TUNE_FILE')
    if extra and os.path.exists(extra):
        lib.ld_conv_tune_load(extra.encode())


def save_tune_table(path):
    return get_lib().ld_conv_tune_save(str(path).encode())


ABI_VERSION = 10
_vp, _i64, _i32, _f32, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_size_t
_G, _H, _M = C.POINTER(GeomT), C.POINTER(LossHpT), C.POINTER(MapsT)
_CV, _EP, _LV = C.POINTER(ConvT), C.POINTER(ConvEpilogueT), C.POINTER(LevelsT)

# name -> (restype, argtypes); kept in one table so the CPU test-suite can
# check that every symbol include/ld_hip.h declares is exported.
SIGNATURES = {
    'ld_abi_version': (C.c_int, []),
    'ld_target_arch': (C.c_char_p, []),
    'ld_atss_targets_workspace_bytes': (_sz, [_G, _i32]),
    'ld_atss_targets': (C.c_int, [_G, _H, _vp, _vp, _vp, _i32, _vp, _vp, _vp,
                                  _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ld_atss_targets_ex': (C.c_int, [_G, _H, _vp, _vp, _vp, _vp, _i32, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _sz, _vp]),
    'ld_grid_anchors': (C.c_int, [_G, _vp, _vp]),
    'ld_loss_workspace_bytes': (_sz, [_G]),
    'ld_fcos_targets': (C.c_int, [_G, _i32, C.POINTER(C.c_float), _i32, _f32,
                                  _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp]),
    'ld_retina_targets_workspace_bytes': (_sz, [_G, _i32, _i32]),
    'ld_retina_targets': (C.c_int, [_G, _i32, _vp, _i32, _f32, _f32, _f32, _i32,
                                    _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ld_loss_prepass_ex': (C.c_int, [_G, _H, _M, _M, _vp, _vp, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _sz, _vp]),
    'ld_loss_prepass': (C.c_int, [_G, _H, _M, _M, _vp, _vp, _vp, _vp, _vp,
                                  _vp, _vp, _sz, _vp]),
    'ld_loss_main': (C.c_int, [_G, _H, _M, _M, _M, _M, _M, _M, _vp, _vp, _vp,
                               _vp, _vp, _vp, _vp, _vp, _vp, _vp, _M, _M, _M,
                               _vp, _sz, _vp]),
    'ld_loss_main_parts': (C.c_int, [_G, _H, _M, _M, _M, _M, _M, _M, _vp, _vp,
                                     _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _M, _M, _M, _M, _M, _M, _vp, _sz, _i32,
                                     _vp]),
    'ld_preprocess_batch': (C.c_int, [_vp, _i32, _i32, _i32,
                                      C.POINTER(C.c_float),
                                      C.POINTER(C.c_float), _i32, _vp, _vp]),
    'ld_deform_im2col': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                   _i32, _i32, _i32, _i32, _vp, _vp]),
    'ld_quality_forward': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp,
                                     _vp, _vp, _vp, _vp]),
    'ld_quality_backward_workspace_bytes': (_sz, [_i32, _i32]),
    'ld_quality_backward': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32,
                                      _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                      _vp, _vp, _i32, _vp, _sz, _vp]),
    'ld_gi_region_workspace_bytes': (_sz, [_G]),
    'ld_gi_region': (C.c_int, [_G, _H, _M, _M, _M, _M, _i32, _f32, _vp, _vp,
                               _vp, _sz, _vp]),
    'ld_loss_finalize': (C.c_int, [_G, _H, _vp, _vp, _vp, _vp, _vp]),
    'ld_loss_centerness': (C.c_int, [_G, _H, _M, _vp, _vp, _vp, _vp, _M, _vp,
                                     _sz, _vp]),
    'ld_loss_set_reg_variant': (C.c_int, [_i32]),
    'ld_probe_copy': (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp]),
    'ld_probe_planes': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _i32, _vp]),
    'ld_gconv_weight_image_floats': (_sz, [_i32, _i32, _i32, _i32]),
    'ld_gconv_weight_transform': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp,
                                            _vp]),
    'ld_gconv_forward': (C.c_int, [_vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32,
                                   _i32, _i32, _i32, _i32, _vp, _vp, _i32,
                                   _vp]),
    'ld_kl_integral_dense': (C.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _vp,
                                       _vp, _vp, _vp]),
    'ld_kd_kl_rows': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _f32, _vp,
                                _vp, _vp]),
    'ld_qfl_rows': (C.c_int, [_vp, _vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp,
                              _vp]),
    'ld_dfl_rows': (C.c_int, [_vp, _vp, _vp, _i64, _i32, _f32, _vp, _vp,
                              _vp]),
    'ld_giou_rows': (C.c_int, [_vp, _vp, _vp, _i64, _f32, _f32, _vp, _vp,
                               _vp]),
    'ld_integral_rows': (C.c_int, [_vp, _i64, _vp, _vp]),
    'ld_integral_rows_bwd': (C.c_int, [_vp, _vp, _i64, _vp, _vp]),
    'ld_bbox_overlaps': (C.c_int, [_vp, _vp, _i64, _i64, _i32, _i32, _f32,
                                   _vp, _vp]),
    'ld_sum': (C.c_int, [_vp, _i64, _vp, _vp, _sz, _vp]),
    'ld_conv_weight_image_floats': (_sz, [_i32, _i32, _i32, _i32, _i32]),
    'ld_conv_weight_transform': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _vp,
                                           _vp, _vp]),
    'ld_get_bboxes_workspace_bytes': (_sz, [_G, _i32, _i32]),
    'ld_get_bboxes': (C.c_int, [_G, _M, _M, _i32, _i32, _vp, _vp, _i32, _f32,
                                _f32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ld_get_bboxes_voting': (C.c_int, [_G, _M, _M, _i32, _i32, _vp, _vp, _i32, _f32,
                                _f32, _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ld_get_bboxes_num_selected': (C.c_int, [_G, _i32, _i32]),
    'ld_get_bboxes_pre_nms': (C.c_int, [_G, _M, _M, _M, _i32, _i32, _i32, _vp,
                                        _vp, _i32, _i32, _vp, _vp, _vp, _vp, _sz,
                                        _vp]),
    'ld_get_bboxes_ex_workspace_bytes': (_sz, [_G, _i32, _i32, _i32]),
    'ld_get_bboxes_ex': (C.c_int, [_G, _M, _M, _M, _i32, _i32, _i32, _vp, _vp, _i32, _f32,
                                   _f32, _i32, _i32, _vp, _vp, _vp, _vp, _sz,
                                   _vp]),
    'ld_conv_weight_transform_batch': (C.c_int, [_vp, _vp, _i32, _vp]),
    'ld_conv_weight_transform_tiles': (C.c_int, [_i32, _i32, _i32, _i32]),
    'ld_conv_weight_transform_batch_tiled': (C.c_int, [_vp, _vp, _i32, _i32,
                                                       _vp]),
    'ld_bn_prepare_batch': (C.c_int, [_vp, _vp, _i32, _vp]),
    'ld_conv_forward': (C.c_int, [_CV, _vp, _vp, _EP, _vp, _vp]),
    'ld_conv_forward_smallc': (C.c_int, [_CV, _vp, _vp, _EP, _vp, _vp]),
    'ld_conv_dgrad': (C.c_int, [_CV, _vp, _vp, _vp, _vp]),
    'ld_conv_dgrad_acc': (C.c_int, [_CV, _vp, _vp, _vp, _vp, _vp]),
    'ld_conv_tune_forward': (C.c_int, [_CV, _vp, _vp, _EP, _vp, _vp]),
    'ld_conv_tune_dgrad': (C.c_int, [_CV, _vp, _vp, _vp, _vp]),
    'ld_conv_bf16_weight_image_elems': (_sz, [_i32, _i32, _i32, _i32, _i32]),
    'ld_conv_bf16_weight_transform': (C.c_int, [_vp, _i32, _i32, _i32, _i32,
                                                _vp, _vp, _vp]),
    'ld_conv_bf16_weight_transform_batch': (C.c_int, [_vp, _vp, _i32, _vp]),
    'ld_conv_bf16_forward': (C.c_int, [_CV, _vp, _vp, _EP, _vp, _vp]),
    'ld_conv_bf16_tune_forward': (C.c_int, [_CV, _vp, _vp, _EP, _vp, _vp]),
    'ld_conv_bf16_dgrad': (C.c_int, [_CV, _vp, _vp, _vp, _vp]),
    'ld_conv_bf16_dgrad_acc': (C.c_int, [_CV, _vp, _vp, _vp, _vp, _vp]),
    'ld_conv_bf16_tune_dgrad': (C.c_int, [_CV, _vp, _vp, _vp, _vp]),
    'ld_conv_bf16_wgrad': (C.c_int, [_CV, _vp, _vp, _vp, _i32, _vp, _sz,
                                     _vp]),
    'ld_conv_bf16_wgrad_c8': (C.c_int, [_CV, _vp, _vp, _vp, _i32, _vp, _sz,
                                        _vp]),
    'ld_conv_to_c8': (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    'ld_conv_bf16_forward_c8': (C.c_int, [_CV, _vp, _vp, _EP, _vp, _vp]),
    'ld_conv_bf16_tune_forward_c8': (C.c_int, [_CV, _vp, _vp, _EP, _vp, _vp]),
    'ld_conv_bf16_dgrad_c8': (C.c_int, [_CV, _vp, _vp, _vp, _vp]),
    'ld_conv_bf16_dgrad_c8_acc': (C.c_int, [_CV, _vp, _vp, _vp, _vp, _vp]),
    'ld_record_begin': (C.c_int, []),
    'ld_record_end': (C.c_int64, []),
    'ld_record_abort': (C.c_int, []),
    'ld_record_count': (C.c_int, [C.c_int64]),
    'ld_record_replay': (C.c_int, [C.c_int64, _vp]),
    'ld_record_free': (C.c_int, [C.c_int64]),
    'ld_stream_fork': (C.c_int, [_vp, _vp]),
    'ld_step_list_build': (C.c_int64, [_vp, _i32]),
    'ld_copy_d2d': (C.c_int, [_vp, _vp, _sz, _vp]),
    'ld_step_list_info': (C.c_int, [C.c_int64, C.POINTER(C.c_int)]),
    'ld_step_list_replay': (C.c_int, [C.c_int64, _vp]),
    'ld_step_list_free': (C.c_int, [C.c_int64]),
    'ld_step_list_last_failure': (C.c_int, [C.POINTER(C.c_int)]),
    'ld_bottleneck_c8_supported': (C.c_int, [_i32, _i32, _i32, _i32]),
    'ld_bottleneck_c8_forward': (C.c_int, [C.POINTER(BottleneckT), _vp, _vp,
                                           _vp]),
    'ld_conv_bf16_tune_dgrad_c8': (C.c_int, [_CV, _vp, _vp, _vp, _vp]),
    'ld_conv_tune_load': (C.c_int, [C.c_char_p]),
    'ld_conv_tune_save': (C.c_int, [C.c_char_p]),
    'ld_conv_tune_clear': (C.c_int, []),
    'ld_conv_wgrad_workspace_bytes': (_sz, [_CV]),
    'ld_conv_wgrad_partial': (C.c_int, [_CV, _i32, _vp, _vp, _vp, _sz,
                                        C.POINTER(WgradJobT), _vp]),
    'ld_wgrad_reduce_batch': (C.c_int, [_vp, _vp, _i32, _vp]),
    'ld_bn_act_backward_nsplit': (C.c_int, [_i32, _i32, _i32, _i32]),
    'ld_bias_grad_nsplit': (C.c_int, [_i32, _i32, _i32]),
    'ld_bias_grad_partial': (C.c_int, [_vp, _i32, _i32, _i32, _vp, _sz, _vp]),
    'ld_bn_bwd_finalize_batch': (C.c_int, [_vp, _vp, _i32, _vp]),
    'ld_conv_tune_wgrad_workspace_bytes': (_sz, [_CV]),
    'ld_conv_wgrad': (C.c_int, [_CV, _vp, _vp, _vp, _i32, _vp, _sz, _vp]),
    'ld_conv_tune_wgrad': (C.c_int, [_CV, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ld_conv_wgrad_plan': (C.c_int, [_CV, C.POINTER(C.c_int)]),
    'ld_bn_prepare': (C.c_int, [_vp, _vp, _vp, _vp, _f32, _i32, _vp, _vp, _vp,
                                _vp]),
    'ld_bn_act_forward_c8': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32,
                                       _i32, _vp, _vp, _vp]),
    'ld_bn_act_forward': (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _i32,
                                    _i32, _vp, _vp]),
    'ld_bn_act_backward_workspace_bytes': (_sz, [_i32, _i32, _i32]),
    'ld_bn_act_backward': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32,
                                     _i32, _i32, _vp, _vp, _vp, _vp, _i32,
                                     _vp, _sz, _vp]),
    'ld_bn_act_backward_c8': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                        _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                                        _vp, _i32, _vp, _sz, _vp]),
    'ld_bn_act_backward_c8in': (C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                          _i32, _i32, _i32, _vp, _vp, _vp, _vp,
                                          _vp, _i32, _vp, _sz, _vp]),
    'ld_bias_grad': (C.c_int, [_vp, _i32, _i32, _i32, _vp, _i32, _vp]),
    'ld_gn_forward_workspace_bytes': (_sz, [_LV, _i32, _i32]),
    'ld_gn_forward': (C.c_int, [_LV, _vp, _vp, _vp, _i32, _i32, _i32, _f32,
                                _i32, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ld_gn_forward_c8': (C.c_int, [_LV, _vp, _vp, _vp, _i32, _i32, _i32, _f32,
                                   _i32, _vp, _vp, _vp, _vp, _vp, _sz, _vp]),
    'ld_gn_backward_c8': (C.c_int, [_LV, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                    _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32,
                                    _vp, _sz, _vp]),
    'ld_gn_backward_c8_lean': (C.c_int, [_LV, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                         _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32,
                                         _vp, _sz, _vp]),
    'ld_gn_backward_workspace_bytes': (_sz, [_LV, _i32, _i32]),
    'ld_gn_backward': (C.c_int, [_LV, _vp, _vp, _vp, _vp, _vp, _vp, _i32,
                                 _i32, _i32, _i32, _vp, _vp, _vp, _i32, _vp,
                                 _sz, _vp]),
    'ld_maxpool3x3s2': (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp]),
    'ld_upsample_add_forward': (C.c_int, [_vp, _vp, _i32, _i32, _i32, _i32,
                                          _i32, _vp, _vp]),
    'ld_upsample_add_backward': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32,
                                           _vp, _vp]),
    'ld_upsample_add_backward_acc': (C.c_int, [_vp, _i32, _i32, _i32, _i32, _i32,
                                               _vp, _vp, _vp]),
    'ld_pack_levels': (C.c_int, [_LV, C.POINTER(C.c_void_p), _i32, _vp, _vp]),
    'ld_unpack_levels': (C.c_int, [_LV, _vp, _i32, C.POINTER(C.c_void_p), _vp]),
    'ld_pack_levels_c8': (C.c_int, [_LV, C.POINTER(C.c_void_p), _i32, _i32, _vp,
                                    _vp, _vp]),
    'ld_unpack_levels_c8': (C.c_int, [_LV, _vp, _i32, _i32,
                                      C.POINTER(C.c_void_p),
                                      C.POINTER(C.c_void_p), _vp]),
    'ld_scale_levels_forward': (C.c_int, [_LV, _vp, _vp, _i32, _vp, _vp]),
    'ld_scale_levels_backward_workspace_bytes': (_sz, [_LV]),
    'ld_scale_levels_backward': (C.c_int, [_LV, _vp, _vp, _vp, _i32, _vp, _vp,
                                           _i32, _vp, _sz, _vp]),
    'ld_sgd_step': (C.c_int, [_vp, _vp, _vp, _sz, _f32, _f32, _f32, _f32,
                              _vp]),
    'ld_sgd_step_dev': (C.c_int, [_vp, _vp, _vp, _sz, _vp, _vp]),
}


def _declare(lib):
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


_ERR = {-1: 'LD_EINVAL (bad argument)', -2: 'LD_ENOSPACE (workspace too small)',
        -3: 'LD_EUNSUPPORTED'}


def check(rc, what):
    if rc != 0:
        raise LdError(f'{what} failed: {_ERR.get(rc, "hipError_t %d" % rc)}')


# ---------------------------------------------------------------------------
# tensor plumbing
# ---------------------------------------------------------------------------
def require_device(t, dtype=None, name='tensor'):
    """The product runs on the GPU only: refuse anything else, loudly."""
    if not isinstance(t, torch.Tensor):
        raise LdError(f'{name}: expected a torch.Tensor, got {type(t)}')
    if not t.is_cuda:
        raise LdError(
            f'{name} is on {t.device}: ld_amd ops only run on a HIP device '
            '(there is no CPU path)')
    if dtype is not None and t.dtype != dtype:
        raise LdError(f'{name}: expected {dtype}, got {t.dtype}')
    return t


# While a launch list is being recorded (ld_record_begin .. ld_record_end, see
# detectors.TeacherPlan) every tensor whose address goes to the library is kept
# alive with the list: a replay re-issues the launches with these very pointers.
_KEEP = [None]


def ptr(t):
    if t is None:
        return C.c_void_p(0)
    k = _KEEP[0]
    if k is not None:
        k.append(t)
    return C.c_void_p(t.data_ptr())


def keep(*ts):
    """For call sites that put ``t.data_ptr()`` into a struct themselves."""
    k = _KEEP[0]
    if k is not None:
        k.extend(t for t in ts if t is not None)


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)


def stream_ptr(device=None):
    """The current HIP stream of ``device`` as a void*.  (torch.cuda.
    current_stream() builds a Python Stream object per call: 4-5 us, ~420 calls
    per step in the round-5 host profile; the raw accessor is a plain C call.)"""
    if isinstance(device, torch.device) and device.type != 'cuda':
        return C.c_void_p(0)
    if _RAW_STREAM is not None:
        if device is None:
            idx = torch.cuda.current_device()
        elif isinstance(device, int):
            idx = device
        else:
            idx = device.index
            if idx is None:
                idx = torch.cuda.current_device()
        return C.c_void_p(_RAW_STREAM(idx))
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def stream_id(device=None):
    """An integer identifying the current stream (cache keys)."""
    return stream_ptr(device).value or 0


def make_geom(featmap_sizes, strides, num_imgs, anchor_scale=8):
    g = GeomT()
    L = len(featmap_sizes)
    if L > LD_MAX_LEVELS or L != len(strides):
        raise LdError('bad pyramid geometry')
    off = 0
    for l, ((h, w), s) in enumerate(zip(featmap_sizes, strides)):
        s = s[0] if isinstance(s, (tuple, list)) else s
        g.lv[l].H, g.lv[l].W = int(h), int(w)
        g.lv[l].stride, g.lv[l].offset = int(s), off
        off += int(h) * int(w)
    g.num_levels, g.num_anchors = L, off
    g.num_imgs, g.anchor_scale = int(num_imgs), int(anchor_scale)
    return g


def make_maps(tensors):
    """Per-level (N, C, H, W) fp32 device tensors -> MapsT.  Each tensor's
    spatial block must be contiguous (stride (.., .., W, 1)); batch and
    channel strides are free, so views into a level-concatenated (N, C, A)
    arena work without copies."""
    m = MapsT()
    for l, t in enumerate(tensors):
        require_device(t, torch.float32, f'level {l} map')
        n, c, h, w = t.shape
        sn, sc, sh, sw = t.stride()
        if (w > 1 and sw != 1) or (h > 1 and sh != w):
            raise LdError(f'level {l} map: spatial block is not contiguous')
        m.ptr[l] = t.data_ptr()
        m.stride_n[l] = sn
        m.stride_c[l] = sc
    return m
