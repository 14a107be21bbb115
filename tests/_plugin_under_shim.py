"""Worker of tests/test_boundary.py::test_reference_side_binding (runs in its
own process: the oracle import shim fabricates mmcv & co. process-wide).

Executes, in the build container, exactly what INTEGRATION.md section 1 tells
a reference maintainer to do: make the *reference's* mmdet importable, import
``ld_amd.mmdet_plugin`` (the custom_imports hook), and build a configs/ld
model through the REFERENCE's own ``mmdet.models.build_detector``.  Prints a
JSON report on stdout."""
import json
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, 'oracle'))


def main():
    cfg_path = sys.argv[1]
    import ref_shim
    ref_shim.install()
    os.chdir(ref_shim.REFERENCE_ROOT)
    import mmdet  # the reference package itself
    assert os.path.realpath(mmdet.__file__).startswith(
        os.path.realpath(ref_shim.REFERENCE_ROOT))
    from mmcv import Config  # shim's restatement of mmcv.Config
    from mmdet.models import build_detector  # reference builder.py:62-64
    from mmdet.models.builder import HEADS
    ref_head = HEADS.get('LDHead')
    assert ref_head.__module__.startswith('mmdet.'), ref_head
    # the reference-side binding: tools/train.py:93-95 would import this module
    # because the config lists it in custom_imports
    import ld_amd.mmdet_plugin  # noqa: F401
    assert HEADS.get('LDHead').__module__.startswith('ld_amd.')
    cfg = Config.fromfile(cfg_path)
    cfg.model['teacher_ckpt'] = None  # URLs cannot be fetched offline
    det = build_detector(cfg.model, train_cfg=cfg.get('train_cfg'),
                         test_cfg=cfg.get('test_cfg'))
    foreign = sorted({type(m).__module__ + '.' + type(m).__name__
                      for mod in (det, det.teacher_model)
                      for m in mod.modules()
                      if not type(m).__module__.startswith(('ld_amd.', 'torch.'))})
    torch_leaf = sorted({type(m).__name__ for mod in (det, det.teacher_model)
                         for m in mod.modules()
                         if type(m).__module__.startswith('torch.')})
    print(json.dumps(dict(
        detector=type(det).__module__ + '.' + type(det).__name__,
        head=type(det.bbox_head).__module__ + '.' + type(det.bbox_head).__name__,
        teacher=type(det.teacher_model).__module__ + '.' +
        type(det.teacher_model).__name__,
        foreign=foreign, torch_containers=torch_leaf,
        student_keys=list(det.state_dict().keys()),
        teacher_keys=list(det.teacher_model.state_dict().keys()),
        trainable=[k for k, p in det.named_parameters() if p.requires_grad])))


if __name__ == '__main__':
    main()
