/* ld_hip.h -- C ABI of libldhip.so, the MI355X (gfx950) implementation of the
 * HikariTJU/LD training hot path.
 *
 * The reference (an MMDetection 2.10 fork, pure Python) has no native
 * boundary of its own: its drop-in boundary is the mmdet registry API
 * (SURVEY.md section 8b).  The host-side mirror of that API lives in the
 * Python package `ld_amd`; every arithmetic step underneath it goes through
 * the entry points declared here.  Each entry point cites the reference
 * function it replaces (file:line relative to the reference checkout).
 *
 * Conventions
 *  - plain pointers and sizes only; all pointers are DEVICE pointers unless the
 *    parameter is a `const ld_*_t*` descriptor struct (host memory, read
 *    during the call) or says "host";
 *  - no allocation, no synchronisation and no host read-back inside: callers
 *    pass workspaces; every launch goes to `stream` (a hipStream_t);
 *  - return 0 on success, a negative LD_E* code on a bad argument, or the
 *    positive hipError_t of a failed launch.  Nothing throws.
 *  - tensors are fp32 unless stated; label tensors are int64 as in the
 *    reference (ld_head.py:522-530).
 */
#ifndef LD_HIP_H_
#define LD_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LD_MAX_LEVELS 8
#define LD_NUM_LOSS_KEYS 8 /* loss_cls,bbox,dfl,ld,ld_vlr,kd,kd_neg,im */

#define LD_EINVAL (-1)   /* bad argument */
#define LD_ENOSPACE (-2) /* workspace too small */
#define LD_EUNSUPPORTED (-3)

typedef void* ld_stream_t; /* hipStream_t */

/* ---- geometry of the FPN pyramid ------------------------------------- */
/* One anchor per feature-map cell, anchors ordered level-major then row-major
 * (x fastest): anchor_generator.py:229-270, ld_head.py:396-403. */
typedef struct {
  int32_t H, W;     /* feature-map size of this level            */
  int32_t stride;   /* anchor stride (8,16,32,64,128)            */
  int32_t offset;   /* index of the level's first anchor         */
} ld_level_t;

typedef struct {
  int32_t num_levels;
  int32_t num_anchors;      /* sum of H*W over levels                */
  int32_t num_imgs;         /* N                                     */
  int32_t anchor_scale;     /* octave_base_scale (8): side = scale*stride */
  ld_level_t lv[LD_MAX_LEVELS];
} ld_geom_t;

/* A per-level family of NCHW maps (N, C, H_l, W_l).  Element (n, c, y, x) of
 * level l lives at ptr[l] + n*stride_n[l] + c*stride_c[l] + y*W_l + x, which
 * covers both separately allocated level tensors and the level-concatenated
 * (N, C, A) arena the head uses. */
typedef struct {
  float* ptr[LD_MAX_LEVELS];
  int64_t stride_n[LD_MAX_LEVELS];
  int64_t stride_c[LD_MAX_LEVELS];
} ld_maps_t;

/* Hyper-parameters of LDHead.loss (ld_head.py:47-71 and the loss modules'
 * constructor arguments). */
typedef struct {
  int32_t num_classes; /* 80 */
  int32_t reg_max;     /* 16 -> 17 bins; only 16 is compiled */
  int32_t topk;        /* ATSS topk, 9 */
  int32_t feat_channels; /* 256 (ld_head.py:153-154 hard-codes it) */
  float lw_cls, qfl_beta;       /* QualityFocalLoss (beta must be 2) */
  float lw_bbox, giou_eps;      /* GIoULoss        */
  float lw_dfl;                 /* DistributionFocalLoss */
  float lw_ld, T_ld;            /* KnowledgeDistillationKLDivLoss (main LD) */
  float lw_ld_vlr, T_ld_vlr;    /* ... (VLR LD)    */
  float lw_kd, T_kd;            /* ... (cls KD)    */
  float lw_im;                  /* IMLoss          */
  /* GFLv2 / LDv2 (gfocal_head.py:201-217, ld_gflv2.py:116-282): the class
   * map is cls_score = sigmoid(cls_feat) * quality, a PROBABILITY with
   * cls_channels = num_classes + 1 channels (use_sigmoid=False,
   * anchor_head.py:68-71).  With LD_LOSS_PROB_CLS: QFL uses
   * binary_cross_entropy on probabilities (gfocal_loss.py:27-30),
   * weight_targets = max_c cls_score without a sigmoid (ld_gflv2.py:200) and
   * the KD term runs on the separate kd_* maps (raw cls_feat, ld_gflv2.py:243).
   * cls_channels = 0 means num_classes. */
  int32_t cls_channels;
  int32_t flags;
  /* LD_LOSS_ATSS (LDATSSHead over ATSSGFLHead, ld_atss.py:44-250): loss_cls is
   * the sigmoid FocalLoss (gamma = qfl_beta = 2, focal_alpha) instead of QFL;
   * the GIoU term is weighted by the ATSS centerness target and normalised by
   * the sum of those targets (1 if that sum is < 1e-12); a centerness BCE term
   * (lw_ctr, ld_loss_centerness) is added and reported in the loss_kd_neg row;
   * there is no DFL term (set lw_dfl = 0) and loss_ld_neg = 0.15 * loss_ld's
   * module on the VLR region (set lw_ld_vlr = 0.6 * lw_ld, T_ld_vlr = T_ld). */
  float lw_ctr, focal_alpha;
} ld_loss_hp_t;
#define LD_LOSS_PROB_CLS 1
/* target assignment: the imitation region is "anchor centre strictly inside a
 * GT box" (get_im_region modes 'fitnet' / 'decouple' / 'gibox',
 * ld_head.py:597-611) instead of the 'finegrained' IoU rule (:594-596) */
#define LD_IM_CENTER_INSIDE 2
#define LD_LOSS_ATSS 4
/* with LD_LOSS_ATSS: LDFCOSHead over FCOSGFLHead (ld_fcos_head.py:46-217).
 * Points instead of anchors: the cell centre is (x + 0.5, y + 0.5) strides,
 * bbox_targets hold (left, top, right, bottom) distances in pixels
 * (ld_fcos_targets), the centerness target comes from those distances, and the
 * VLR term is weighted by vlr * max_c sigmoid(cls) (the "remain" points,
 * ld_fcos_head.py:117-131; set lw_ld_vlr = lw_ld for its 0.25 factor). */
#define LD_LOSS_FCOS 8
/* LDRetinaHead over RetinaGFLHead (ld_retina.py:41-137), B anchors per cell:
 * the caller views its (N, B * C, H, W) maps as (N * B, C, H, W) -- N * B
 * pseudo-images with one anchor per cell -- and passes geom.num_imgs = N * B
 * with the target arrays of ld_retina_targets.  FocalLoss (focal_alpha) on
 * every anchor with label_weights (0 in the ignore band); GIoU with weight 1
 * on the positives; loss_cls and loss_bbox divided by num_total_pos
 * (counts[num_imgs + 2L], LOCAL: the head takes no cross-rank mean); LD and
 * the VLR term take the softmax over ALL 68 corner logits of an anchor (the
 * head hands (rows, 68) to the KL loss, ld_retina.py:87-110), weights max_c
 * sigmoid(cls) on the positives / the VLR value on background anchors, both
 * with avg_factor 4 (set lw_ld_vlr = 0.03 * 4 * lw_ld, T_ld_vlr = T_ld); no DFL
 * (lw_dfl = 0); KD on the positives' class logits as in LDHead. */
#define LD_LOSS_RETINA 16

/* ---- library ------------------------------------------------------------ */
/* ABI version of this header; bump on any signature change. */
int ld_abi_version(void);
/* Name of the gfx target the kernels were compiled for ("gfx950"). */
const char* ld_target_arch(void);

/* ---- target assignment ---------------------------------------------------
 * Replaces, for a whole batch in two launches:
 *   AnchorHead.get_anchors            anchor_head.py:145-173 (anchors implicit)
 *   anchor_inside_flags               core/anchor/utils.py:20-46
 *   ATSSAssigner.assign               atss_assigner.py:33-181
 *   PseudoSampler.sample              samplers/pseudo_sampler.py:24-41
 *   ATSSAssigner.get_vlr_region       atss_assigner.py:183-298
 *   LDHead.get_im_region('finegrained') ld_head.py:580-611
 *   LDHead._get_target_single/unmap   ld_head.py:449-577
 *
 * gt_bboxes  (N, max_gt, 4) xyxy, rows >= num_gt[n] ignored
 * gt_labels  (N, max_gt) int64
 * num_gt     (N) int32
 * valid_hw   (N, num_levels, 2) int32: valid_h, valid_w of each level from
 *            img_meta['pad_shape'] (anchor_generator.py:293-300)
 * outputs, dense (N, A): labels int64 (background = num_classes),
 *   label_weights, bbox_targets (N, A, 4), vlr, im
 * counts     int32 [N + 2*num_levels + 1]:
 *            [0,N) positives per image; [N, N+L) positives per level (batch);
 *            [N+L, N+2L) im-region anchors per level; [N+2L] = sum_i max(P_i,1)
 * Tie rule: equal centre distances are ordered by lower anchor index (the
 * reference's order is an artefact of torch.topk). */
size_t ld_atss_targets_workspace_bytes(const ld_geom_t* geom, int max_gt);
int ld_atss_targets(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                    const float* gt_bboxes, const int64_t* gt_labels,
                    const int32_t* num_gt, int max_gt, const int32_t* valid_hw,
                    int64_t* labels, float* label_weights, float* bbox_targets,
                    float* vlr, float* im, int32_t* counts, void* workspace,
                    size_t workspace_bytes, ld_stream_t stream);

/* Same, with (a) optional EXPLICIT anchors (A, 4) -- the reference-style
 * ATSSAssigner.assign(bboxes, num_level_bboxes, ...) entry, where a level is
 * just an index range: pass geometry levels with H = count, W = 1 -- and
 * (b) optional per-anchor assign results: gt_inds (N, A) int64 1-based / 0,
 * max_overlaps (N, A) (-1e8 where unassigned) as AssignResult carries them
 * (atss_assigner.py:163-181).  anchors / gt_inds / max_overlaps may be NULL. */
int ld_atss_targets_ex(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                       const float* anchors, const float* gt_bboxes,
                       const int64_t* gt_labels, const int32_t* num_gt,
                       int max_gt, const int32_t* valid_hw, int64_t* labels,
                       float* label_weights, float* bbox_targets, float* vlr,
                       float* im, int32_t* counts, int64_t* gt_inds,
                       float* max_overlaps, void* workspace,
                       size_t workspace_bytes, ld_stream_t stream);

/* Materialise the anchors (A, 4) -- only for API compatibility
 * (AnchorGenerator.grid_anchors, anchor_generator.py:207-270); the kernels
 * never read them. */
/* FCOS point targets for the whole batch (LDFCOSHead.get_targets,
 * ld_fcos_head.py:261-414): same output arrays as ld_atss_targets; bbox_targets
 * = (l, t, r, b) distances of the assigned object, vlr = 1 on the points that
 * lie inside some gt box but are assigned to none ("remain"), label_weights = 1,
 * im = 0.  regress_ranges: HOST float[2 * num_levels] (lo, hi per level).
 * counts[N + 2L] = total number of positives (no per-image floor). */
int ld_fcos_targets(const ld_geom_t* geom, int num_classes,
                    const float* regress_ranges, int center_sampling,
                    float center_sample_radius, const float* gt_bboxes,
                    const int64_t* gt_labels, const int32_t* num_gt, int max_gt,
                    int64_t* labels, float* label_weights, float* bbox_targets,
                    float* vlr, float* im, int32_t* counts, ld_stream_t stream);
int ld_grid_anchors(const ld_geom_t* geom, float* anchors, ld_stream_t stream);

/* MaxIoU targets for a head with ``num_base`` anchors per cell
 * (LDRetinaHead.get_targets / _get_targets_single, ld_retina.py:256-470):
 * MaxIoUAssigner.assign (max_iou_assigner.py:93-212; match_low_quality and
 * gt_max_assign_all on, no ignore boxes, float neg_iou_thr) + PseudoSampler +
 * the head's get_vlr_region (ld_retina.py:478-603).  ``geom`` describes the N
 * real images and the H x W cells; ``anchors`` is the explicit list
 * (num_anchors * num_base, 4) in AnchorGenerator.grid_anchors order (level,
 * cell, base anchor fastest).  Outputs use the PSEUDO-IMAGE layout
 * [(n * num_base + b) * num_anchors + cell] (see LD_LOSS_RETINA): labels
 * (background = num_classes), label_weights (1 on positives and negatives, 0 in
 * the band between the thresholds and outside the valid region), bbox_targets
 * (the matched gt box, xyxy pixels), vlr, im (zeros), optional gt_inds (-1
 * ignored / 0 negative / g + 1).  counts: int32[N * num_base + 2L + 1]: positives
 * per pseudo-image, per level, (unused), and at [N * num_base + 2L] the
 * reference's num_total_pos = sum over REAL images of max(P_i, 1). */
size_t ld_retina_targets_workspace_bytes(const ld_geom_t* geom, int num_base,
                                         int max_gt);
int ld_retina_targets(const ld_geom_t* geom, int num_base, const float* anchors,
                      int num_classes, float pos_iou_thr, float neg_iou_thr,
                      float min_pos_iou, int topk, const float* gt_bboxes,
                      const int64_t* gt_labels, const int32_t* num_gt, int max_gt,
                      const int32_t* valid_hw, int64_t* labels,
                      float* label_weights, float* bbox_targets, float* vlr,
                      float* im, int32_t* counts, int64_t* gt_inds,
                      void* workspace, size_t workspace_bytes, ld_stream_t stream);

/* ---- fused loss block ----------------------------------------------------
 * Replaces LDHead.loss_single for all levels and images, forward AND gradient
 * (ld_head.py:116-282), reading the head outputs NCHW-direct (no permute).
 *
 * Call order on one stream:
 *   ld_loss_prepass   -> weight_targets, IoU score, partial normalisers
 *   (optional cross-rank all-reduce of norm_partial[0..1])
 *   ld_loss_main      -> loss partials + gradients
 *   ld_loss_finalize  -> the (8, L) loss table
 *
 * norm (device, float[4]): [0] sum_i max(P_i,1) (this rank), [1] sum of
 *   weight_targets (this rank); after the caller's optional mean over ranks
 *   ld_loss_main reads NTS = max(norm[0],1) and AVG = norm[1] + 1e-6 from it
 *   (ld_head.py:338-341,362-365).
 */
size_t ld_loss_workspace_bytes(const ld_geom_t* geom);

/* weight_targets = max_c sigmoid(cls) at positives (ld_head.py:198-199),
 * score = IoU(decoded box, target) (ld_head.py:200-207) written dense (N, A)
 * (zero elsewhere); norm[0], norm[1] as above. */
int ld_loss_prepass(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                    const ld_maps_t* cls, const ld_maps_t* reg,
                    const int64_t* labels, const float* bbox_targets,
                    const int32_t* counts, float* weight_targets, float* score,
                    float* norm, void* workspace, size_t workspace_bytes,
                    ld_stream_t stream);

/* The same with the VLR array: needed by LD_LOSS_FCOS (weight_targets are also
 * produced where vlr > 0); vlr may be NULL otherwise. */
int ld_loss_prepass_ex(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                       const ld_maps_t* cls, const ld_maps_t* reg,
                       const int64_t* labels, const float* bbox_targets,
                       const float* vlr, const int32_t* counts,
                       float* weight_targets, float* score, float* norm,
                       void* workspace, size_t workspace_bytes, ld_stream_t stream);

/* Forward + gradient of every loss term.  grad_* have the same descriptors
 * as their inputs and are fully overwritten.  `upstream` (device,
 * float[8 * num_levels] key-major like `losses`, may be NULL = all ones) is
 * d total / d loss[key][level] (base.py:197-210 sums them, so it is 1). */
int ld_loss_main(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                 const ld_maps_t* cls, const ld_maps_t* reg,
                 const ld_maps_t* t_cls, const ld_maps_t* t_reg,
                 const ld_maps_t* x, const ld_maps_t* t_x,
                 const int64_t* labels, const float* label_weights,
                 const float* bbox_targets, const float* vlr, const float* im,
                 const int32_t* counts, const float* weight_targets,
                 const float* score, const float* norm, const float* upstream,
                 const ld_maps_t* grad_cls, const ld_maps_t* grad_reg,
                 const ld_maps_t* grad_x, void* workspace,
                 size_t workspace_bytes, ld_stream_t stream);

/* The same with a launch mask: ld_loss_main runs four dense launches --
 * POS (the positive anchors: GIoU + class-softmax statistics), REG (thread =
 * anchor x side: LD-KL + VLR-LD + DFL + the GIoU chain through Integral,
 * forward and gradient -- the north-star fused LD-KL + Integral sweep), CLS
 * (anchor x 16 classes: QFL + KD), IM (anchor x 32 channels: masked MSE).
 * `parts` selects which of them run (benchmarking one kernel of the step at a
 * saturating size; REG/CLS read what POS wrote for the positive anchors).
 * kd_s / kd_t / grad_kd: student / teacher maps of the KD term and its
 * gradient map when they are not the class maps themselves (LDv2: raw
 * cls_feat); NULL = cls / t_cls / grad_cls (LDHead: KD adds into grad_cls). */
#define LD_LOSS_PART_POS 1
#define LD_LOSS_PART_REG 2
#define LD_LOSS_PART_CLS 4
#define LD_LOSS_PART_IM 8
#define LD_LOSS_PART_ALL 15
/* Selects the kernel the REG part launches (benchmarking / A-B tests; the
 * default is the measured best, profiles/r03_ldkl_variants.json).  variant:
 * bits 0-1 log2(anchors per thread: 1, 2, 4) | bit 2 non-temporal loads | bit 3
 * non-temporal stores | bit 4 hardware exp / rcp | bit 5 the four sides of an
 * anchor chunk in adjacent workgroups | bit 6 the NT bits also apply to maps
 * that fit the Infinity Cache | bit 7 (1 anchor per thread only) hold the
 * kernel to 64 VGPRs = 8 waves per SIMD | bit 16 (with bit 5) those four
 * workgroups on one XCD | bits 8-15 KiB of dynamic LDS per workgroup
 * (occupancy throttle); < 0 = the round-2 kernel (which LD_LOSS_RETINA and
 * T_ld != T_ld_vlr always use).  Returns the previous value (-1 if it was
 * negative), LD_EINVAL on a malformed word.  Not thread-safe; set it between
 * launches. */
int ld_loss_set_reg_variant(int variant);
/* LD_LOSS_ATSS only: the centerness term.  ctr / grad_ctr: (N, 1, H_l, W_l)
 * maps; score = the centerness targets ld_loss_prepass wrote; call between
 * ld_loss_main_parts and ld_loss_finalize (same workspace). */
int ld_loss_centerness(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                       const ld_maps_t* ctr, const int64_t* labels,
                       const float* score, const float* norm, const float* upstream,
                       const ld_maps_t* grad_ctr, void* workspace,
                       size_t workspace_bytes, ld_stream_t stream);
int ld_loss_main_parts(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                       const ld_maps_t* cls, const ld_maps_t* reg,
                       const ld_maps_t* t_cls, const ld_maps_t* t_reg,
                       const ld_maps_t* x, const ld_maps_t* t_x,
                       const int64_t* labels, const float* label_weights,
                       const float* bbox_targets, const float* vlr, const float* im,
                       const int32_t* counts, const float* weight_targets,
                       const float* score, const float* norm, const float* upstream,
                       const ld_maps_t* grad_cls, const ld_maps_t* grad_reg,
                       const ld_maps_t* grad_x, const ld_maps_t* kd_s,
                       const ld_maps_t* kd_t, const ld_maps_t* grad_kd,
                       void* workspace, size_t workspace_bytes, int parts,
                       ld_stream_t stream);

/* The 'gibox' imitation region (LDHead.get_gi_region, ld_head.py:613-637;
 * LDv2Head: ld_gflv2.py:619-644 with hp->flags & LD_LOSS_PROB_CLS): per level,
 * over all cells of all images, the General-Instance score max_c |teacher -
 * student| and box (the winner's decoded distribution), then the first `topn`
 * (10) survivors of a greedy NMS at `iou_thr` (0.3) -- torchvision.ops.nms's
 * published semantics: descending score, IoU > thr suppresses.  Writes the
 * (N, A) 0/1 mask `im` and the per-level selected counts into
 * counts[N + L + l] (the slots ld_loss_main reads for loss_im); run it between
 * ld_atss_targets and ld_loss_main.  Equal scores: lower cell index first. */
size_t ld_gi_region_workspace_bytes(const ld_geom_t* geom);
int ld_gi_region(const ld_geom_t* geom, const ld_loss_hp_t* hp, const ld_maps_t* cls,
                 const ld_maps_t* reg, const ld_maps_t* t_cls, const ld_maps_t* t_reg,
                 int topn, float iou_thr, float* im, int32_t* counts, void* workspace,
                 size_t workspace_bytes, ld_stream_t stream);

/* losses: device float[8 * num_levels], key-major (loss_cls[0..L), ...). */
int ld_loss_finalize(const ld_geom_t* geom, const ld_loss_hp_t* hp,
                     const int32_t* counts, const float* norm,
                     const void* workspace, float* losses, ld_stream_t stream);

/* The north-star kernel on its own, at any size: LD KL (T) + Integral
 * (+ fused gradient) over `rows`-many anchors x 4 sides, channel-major
 * (68, rows) student and teacher logits, weight (rows).
 *   integral (4, rows) out, loss_rows (4, rows) out = weight * KL per side,
 *   grad (68, rows) out or NULL (forward only).
 * Algorithmic bytes per anchor-side row: 136 logits in + 4 weight + 4 integral
 * + 4 loss (+ 68 gradient) = 148 forward / 216 fused forward+gradient.
 * Replaces Integral.forward (gfl_head.py:32-44) +
 * knowledge_distillation_kl_div_loss (kd_loss.py:10-36) on the dense map. */
int ld_kl_integral_dense(const float* s_reg, const float* t_reg,
                         const float* weight, int64_t rows, float T,
                         float scale, float* integral, float* loss_rows,
                         float* grad, ld_stream_t stream);

/* ---- reference-layout loss operators (row-major (rows, K) tensors) -------
 * These back the registry loss modules when they are called on their own,
 * with the reference's tensor layout.  Each computes the per-row loss
 * (already multiplied by weight[row] if weight != NULL) and, if grad != NULL,
 * d(sum_rows loss_rows * gscale)/d pred. */
/* knowledge_distillation_kl_div_loss, kd_loss.py:10-36 */
int ld_kd_kl_rows(const float* pred, const float* soft, const float* weight,
                  int64_t rows, int K, float T, float gscale, float* loss_rows,
                  float* grad, ld_stream_t stream);
/* quality_focal_loss, gfocal_loss.py:8-50 (beta = 2) */
int ld_qfl_rows(const float* pred, const int64_t* label, const float* score,
                const float* weight, int64_t rows, int C, float gscale,
                float* loss_rows, float* grad, ld_stream_t stream);
/* distribution_focal_loss, gfocal_loss.py:53-74 */
int ld_dfl_rows(const float* pred, const float* target, const float* weight,
                int64_t rows, int K, float gscale, float* loss_rows,
                float* grad, ld_stream_t stream);
/* giou_loss, iou_loss.py:85-102 (aligned boxes (rows, 4)) */
int ld_giou_rows(const float* pred, const float* target, const float* weight,
                 int64_t rows, float eps, float gscale, float* loss_rows,
                 float* grad, ld_stream_t stream);
/* Integral.forward, gfl_head.py:32-44: (rows, 4*(reg_max+1)) -> (rows, 4);
 * backward: grad_in (rows,4) -> grad (rows, 68) */
int ld_integral_rows(const float* x, int64_t rows, float* out,
                     ld_stream_t stream);
int ld_integral_rows_bwd(const float* x, const float* grad_out, int64_t rows,
                         float* grad_x, ld_stream_t stream);
/* bbox_overlaps, iou2d_calculator.py:43-188; mode 0 iou, 1 iof, 2 giou,
 * 3 diou; pairwise (m, n) or aligned (m). */
int ld_bbox_overlaps(const float* b1, const float* b2, int64_t m, int64_t n,
                     int mode, int aligned, float eps, float* out,
                     ld_stream_t stream);
/* sum of a float vector into out[0] (deterministic two-stage) */
int ld_sum(const float* x, int64_t n, float* out, void* workspace,
           size_t workspace_bytes, ld_stream_t stream);

/* ---- convolution (fp32, implicit GEMM on the f32 matrix cores) -----------
 * Replaces nn.Conv2d under ResNet / FPN / GFLHead:
 *   backbones/resnet.py:35-46,163-183 (BasicBlock/Bottleneck convs),
 *   models/utils/res_layer.py:38-59 (downsample), necks/fpn.py:121-160,
 *   dense_heads/gfl_head.py:102-133.
 * Activations are (N, C, P) fp32 with P = sum over levels of H_l*W_l: a plain
 * NCHW tensor is the one-level case; the FPN/head use the level-concatenated
 * form so weight-shared convs over all five levels are one launch. */
typedef struct {
  int32_t Hin, Win, Hout, Wout;
  int32_t off_in, off_out; /* first position of the level inside Pin / Pout */
} ld_conv_level_t;

typedef struct {
  int32_t N, Cin, Cout, KH, KW;
  int32_t stride; /* 1 or 2 */
  int32_t pad;
  int32_t num_levels;
  int32_t Pin, Pout;
  ld_conv_level_t lv[LD_MAX_LEVELS];
} ld_conv_t;

/* Fused forward epilogue, applied in this order:
 *   v = acc; if scale: v = v*scale[co] + shift[co]  (frozen/eval BatchNorm,
 *   resnet.py:639-648); if bias: v += bias[co]; if residual: v += residual;
 *   if relu: v = max(v, 0).  Any pointer may be NULL.  (With y_raw: scale and
 *   bias together are not supported -- a conv followed by a norm has no bias.) */
typedef struct {
  const float* bias;
  const float* scale;
  const float* shift;
  const float* residual; /* same shape as y */
  int32_t relu;
  int32_t reserved;
  /* bf16 forward entry points only (NULL elsewhere): also write the final values
   * as the bf16 channel-blocked image (N, Cout/8, Pout, 8) of ld_conv_to_c8 --
   * bit-identical to converting y afterwards -- for the conv that consumes y
   * next.  Needs Cout % 8 == 0. */
  void* y_c8;
  /* bf16 forward entry points only: the residual as a C8 image instead of the
   * fp32 `residual` (give one or the other), and -- when y_c8 is given -- y
   * itself may be NULL: the frozen teacher then keeps ONLY the bf16 image of its
   * activations (the reference's mixed-precision nets are fp16 end to end the
   * same way, mmcv auto_fp16). */
  const void* residual_c8;
  /* Second output, same shape as y (forward entry points, MODE 0): the conv
   * result BEFORE scale / shift / residual / relu (acc + bias).  A trainable
   * conv -> eval-BN -> ReLU pair then runs as ONE launch: y is what the next
   * layer reads, y_raw is what the BN backward needs for d(gamma)
   * (resnet.py:639-648 keeps BN in eval mode; its affine stays trainable).
   * NULL = not written. */
  float* y_raw;
  /* ld_conv_bf16_forward_c8 only: that second output as a bf16 C8 image
   * (N, Cout/8, Pout, 8) instead of fp32 (give one or the other) -- what
   * ld_bn_act_backward_c8in reads for d(gamma); half the bytes (round 6). */
  void* y_raw_c8;
} ld_conv_epilogue_t;

/* (Cout,Cin,KH,KW) parameter -> GEMM images: wt_fwd [tap][Cin_pad][Cout]
 * and/or wt_bwd [KH*KW-1-tap][Cout_pad][Cin] (either may be NULL); *_pad = the
 * reduction extent rounded up to 32 rows, zero filled.  Image sizes in floats
 * come from ld_conv_weight_image_floats(..., backward = 0 | 1). */
size_t ld_conv_weight_image_floats(int Cout, int Cin, int KH, int KW,
                                   int backward);
int ld_conv_weight_transform(const float* w, int Cout, int Cin, int KH, int KW,
                             float* wt_fwd, float* wt_bwd, ld_stream_t stream);
/* Many weight images in ONE launch (after the optimizer step every trainable
 * conv of the student needs new images: 137 launches of a few microseconds
 * each otherwise).  `jobs` and `block_job` live in DEVICE memory; job j owns
 * the 256-thread blocks [first_block, first_block of job j+1); block_job[b] is
 * the job of block b.  wt_fwd / wt_bwd may be NULL per job. */
typedef struct {
  const float* w;
  float* wt_fwd;
  float* wt_bwd;
  int32_t Cout, Cin, ntaps, first_block;
} ld_wt_job_t;
int ld_conv_weight_transform_batch(const ld_wt_job_t* jobs, const int32_t* block_job,
                                   int nblocks, ld_stream_t stream);
/* The same images from 32 x 32 x ntaps channel tiles staged through LDS (round 5:
 * coalesced reads AND writes; the per-element kernels read the parameter with a
 * stride of Cin * ntaps floats -- 250 us per step for the R50 student).  A job owns
 * ld_conv_weight_transform_tiles(Cout, Cin, ntaps, bf16) consecutive blocks (0:
 * more than 9 taps, not served); bf16 != 0 writes the bf16 images of
 * ld_conv_bf16_weight_transform (wt_fwd / wt_bwd of the job then point at those).
 * Bit-identical to the per-element kernels. */
int ld_conv_weight_transform_tiles(int Cout, int Cin, int ntaps, int bf16);
int ld_conv_weight_transform_batch_tiled(const ld_wt_job_t* jobs,
                                         const int32_t* block_job, int nblocks, int bf16,
                                         ld_stream_t stream);
/* Enqueue only: no timing, no synchronisation, capturable into a hipGraph.
 * The tile shape of a launch comes from the tuning table (below) and, for a
 * geometry the table does not hold, from a model that is a pure function of the
 * geometry -- so two processes with the same table produce the same bits (the
 * KS = 4 split-K shapes sum in a different order than KS = 1). */
int ld_conv_forward(const ld_conv_t* c, const float* x, const float* wt_fwd,
                    const ld_conv_epilogue_t* ep, float* y, ld_stream_t stream);
/* dx (N,Cin,Pin) fully overwritten. */
int ld_conv_dgrad(const ld_conv_t* c, const float* dy, const float* wt_bwd,
                  float* dx, ld_stream_t stream);
/* dx = conv_transpose(dy) + addend, the sum formed in the GEMM epilogue.  addend
 * is (N,Cin,Pin) fp32 and may be dx itself.  Replaces the separate elementwise
 * add autograd issues wherever an activation has two consumers: the block input
 * of a residual block (mmdet/models/backbones/resnet.py:260-299: `out +=
 * identity`), a backbone stage output read by the next stage and the FPN lateral
 * (necks/fpn.py:170-176), an FPN level read by both head towers
 * (dense_heads/gfl_head.py:164-172).  Same tile shape (shape-table record) as
 * ld_conv_dgrad.  ld_conv_bf16_dgrad_acc / ld_conv_bf16_dgrad_c8_acc: the bf16
 * families. */
int ld_conv_dgrad_acc(const ld_conv_t* c, const float* dy, const float* wt_bwd,
                      const float* addend, float* dx, ld_stream_t stream);
/* ---- shape tuning (explicit; cudnn.benchmark's role, kept OUT of the launch
 * path).  ld_conv_tune_* time every candidate tile shape of the geometry by
 * launching the (idempotent) kernel on the caller's buffers -- y / dx must not
 * alias an input -- and record the winner: they synchronise the device and
 * must not be called on a capturing stream.  Return 0 = tuned now, 1 = the
 * geometry was already in the table (nothing done), < 0 / hipError_t on error.
 * ld_conv_tune_load merges a text table (one record per line: 18 key ints,
 * then tm tn wvm d ks; '#' comments) and returns the number of records;
 * ld_conv_tune_save writes the whole table; host paths. */
int ld_conv_tune_forward(const ld_conv_t* c, const float* x, const float* wt_fwd,
                         const ld_conv_epilogue_t* ep, float* y,
                         ld_stream_t stream);
int ld_conv_tune_dgrad(const ld_conv_t* c, const float* dy, const float* wt_bwd,
                       float* dx, ld_stream_t stream);
int ld_conv_tune_load(const char* path);
int ld_conv_tune_save(const char* path);
int ld_conv_tune_clear(void);
/* Workspace of the plan a launch of this geometry actually follows (one size for
 * ld_conv_wgrad / ld_conv_bf16_wgrad / ld_conv_bf16_wgrad_c8).
 * ld_conv_tune_wgrad_workspace_bytes: the worst case over every candidate
 * ld_conv_tune_wgrad times (<= 128 MB), needed by that call only. */
size_t ld_conv_wgrad_workspace_bytes(const ld_conv_t* c);
size_t ld_conv_tune_wgrad_workspace_bytes(const ld_conv_t* c);
/* dw (Cout,Cin,KH,KW): overwritten, or += if accumulate != 0.  Deterministic:
 * the j-reduction is split in a fixed pattern (per wave, per k-group, per
 * workgroup) and every partial sum is combined in index order -- by a second
 * launch, or inside the launch by the last-arriving workgroup of a tile (no
 * float atomics either way).  Launches that may run concurrently must not share
 * a workspace. */
int ld_conv_wgrad(const ld_conv_t* c, const float* x, const float* dy, float* dw,
                  int accumulate, void* workspace, size_t workspace_bytes,
                  ld_stream_t stream);
/* Deferred reduction (round 5): a parameter gradient is needed when its bucket is
 * reduced across ranks / the optimizer runs, not when its layer's backward runs.
 * ld_conv_wgrad_partial computes only the split partials of a weight gradient --
 * family 0 = the fp32 kernels of ld_conv_wgrad, 1 = ld_conv_bf16_wgrad, 2 =
 * ld_conv_bf16_wgrad_c8 (x / dy as C8 images) -- into `slabs` (>=
 * ld_conv_wgrad_workspace_bytes(c), owned by this weight until the batch ran) and
 * fills job->slabs / splits / ntaps / Cout / Cin; the caller sets dw, accumulate
 * and first_block and uploads the jobs of a bucket once.  ld_wgrad_reduce_batch
 * sums every job of a table in ONE launch (block b serves 1024 consecutive
 * [tap][co][ci] elements of job block_job[b]; a job owns ceil(ntaps*Cout*Cin /
 * 1024) consecutive blocks from first_block): the same additions in the same
 * order as the per-layer reduce launch it replaces, 53 launches per step fewer. */
typedef struct {
  const float* slabs; /* [splits][ntaps][Cout][Cin] */
  float* dw;          /* (Cout, Cin, KH, KW) */
  int32_t splits, ntaps, Cout, Cin, accumulate, first_block;
} ld_wgrad_job_t;
int ld_conv_wgrad_partial(const ld_conv_t* c, int family, const void* x, const void* dy,
                          void* slabs, size_t slab_bytes, ld_wgrad_job_t* job,
                          ld_stream_t stream);
int ld_wgrad_reduce_batch(const ld_wgrad_job_t* jobs, const int32_t* block_job,
                          int nblocks, ld_stream_t stream);
/* The plan ld_conv_wgrad would follow for this geometry, without touching the
 * device: out[0..4] = kind (0 wave-private 64 x 64 tiles, 1 workgroup tiles, 2
 * three kw taps per workgroup), k-groups, columns per slice, splits along the
 * reduction, in-launch combination.  Order of precedence: LD_CONV_WGRAD_CFG, the
 * shape table (MODE 2), a model that is a pure function of the geometry. */
int ld_conv_wgrad_plan(const ld_conv_t* c, int* out);
/* Times the fp32 weight-gradient kernels / split shapes of this geometry on the
 * caller's buffers (dw is overwritten) and records the winner in the shape
 * table (key MODE 2).  Same rules and return values as ld_conv_tune_forward;
 * workspace_bytes >= ld_conv_tune_wgrad_workspace_bytes(c). */
int ld_conv_tune_wgrad(const ld_conv_t* c, const float* x, const float* dy, float* dw,
                       void* workspace, size_t workspace_bytes, ld_stream_t stream);

/* ---- convolution, bf16 matrix operands (BASELINE.json config 3) -----------
 * The same three GEMMs on v_mfma_f32_32x32x16_bf16 (16x the f32-MFMA rate).
 * Counterpart of the reference's mixed-precision mode for these layers (mmcv
 * auto_fp16 around ResNet / FPN / GFLHead forward; the reg output is cast back
 * by .float(), gfl_head.py:181-183; the loss block is @force_fp32,
 * ld_head.py:284).  Contract: x / dy / y / dx / dw, the epilogue operands and
 * the accumulators are fp32 exactly as in the fp32 entry points; only the two
 * MFMA operands are rounded to bf16 (round-to-nearest-even) on the way in.
 * Weight images: wt_fwd bf16 [tap][Cin16/8][Cout][8], wt_bwd bf16
 * [KH*KW-1-tap][Cout16/8][Cin][8] (C16 = channels rounded up to 16, zero
 * filled), sizes in bf16 ELEMENTS from ld_conv_bf16_weight_image_elems.
 * forward needs Cin % 16 == 0, dgrad Cout % 16 == 0 (LD_EUNSUPPORTED otherwise:
 * use the fp32 entry point); wgrad takes any channel counts and the workspace of
 * ld_conv_wgrad_workspace_bytes.  ld_conv_bf16_tune_* = ld_conv_tune_* for the
 * bf16 kernel family (same table, family field 1). */
size_t ld_conv_bf16_weight_image_elems(int Cout, int Cin, int KH, int KW,
                                       int backward);
int ld_conv_bf16_weight_transform(const float* w, int Cout, int Cin, int KH, int KW,
                                  void* wt_fwd, void* wt_bwd, ld_stream_t stream);
int ld_conv_bf16_weight_transform_batch(const ld_wt_job_t* jobs,
                                        const int32_t* block_job, int nblocks,
                                        ld_stream_t stream);
int ld_conv_bf16_forward(const ld_conv_t* c, const float* x, const void* wt_fwd,
                         const ld_conv_epilogue_t* ep, float* y, ld_stream_t stream);
int ld_conv_bf16_tune_forward(const ld_conv_t* c, const float* x, const void* wt_fwd,
                              const ld_conv_epilogue_t* ep, float* y,
                              ld_stream_t stream);
int ld_conv_bf16_dgrad(const ld_conv_t* c, const float* dy, const void* wt_bwd,
                       float* dx, ld_stream_t stream);
int ld_conv_bf16_tune_dgrad(const ld_conv_t* c, const float* dy, const void* wt_bwd,
                            float* dx, ld_stream_t stream);
int ld_conv_bf16_wgrad(const ld_conv_t* c, const float* x, const float* dy, float* dw,
                       int accumulate, void* workspace, size_t workspace_bytes,
                       ld_stream_t stream);
/* bf16 channel-blocked ("C8") activation operand.  ld_conv_to_c8 rewrites an
 * fp32 (N, C, P) tensor as bf16 (N, C/8, P, 8) (round to nearest even; C a
 * multiple of 8; `out` holds N*C*P bf16).  The *_c8 entry points take that
 * image as the activation operand of the forward conv (x_c8, Cin % 32 == 0) or
 * of the data gradient (dy_c8, Cout % 32 == 0) -- one 16-byte load per
 * (position, 8 channels), no conversion inside the GEMM loop; weights, the
 * fused epilogue and the fp32 (N, C, P) output are those of
 * ld_conv_bf16_forward / ld_conv_bf16_dgrad, and so are the results bit for
 * bit (the same bf16 operands enter the same fp32 accumulation order as the
 * LDS-tiled kernel of the fp32-input path).  LD_EUNSUPPORTED when the channel
 * count does not fit: use the fp32-input entry point. */
int ld_conv_to_c8(const float* x, int N, int C, int P, void* out, ld_stream_t stream);
int ld_conv_bf16_forward_c8(const ld_conv_t* c, const void* x_c8, const void* wt_fwd,
                            const ld_conv_epilogue_t* ep, float* y,
                            ld_stream_t stream);
int ld_conv_bf16_tune_forward_c8(const ld_conv_t* c, const void* x_c8,
                                 const void* wt_fwd, const ld_conv_epilogue_t* ep,
                                 float* y, ld_stream_t stream);
int ld_conv_bf16_dgrad_c8(const ld_conv_t* c, const void* dy_c8, const void* wt_bwd,
                          float* dx, ld_stream_t stream);
/* ---- launch lists ------------------------------------------------------------
 * Between ld_record_begin and ld_record_end every launch the calling thread makes
 * through this library is executed AND appended (kernel, grid, by-value copy of
 * its arguments) to a list; ld_record_replay re-issues the list on any stream with
 * one C loop -- no shape-table lookup, no descriptor building, no host language in
 * the per-launch path.  Everything a recorded launch points at (activations,
 * weight images, workspaces) must stay allocated while the list is used; tuning
 * entry points must not be called while recording.  ld_record_end returns a handle
 * > 0 (or a negative error); ld_record_count = launches in the list.  Used for the
 * frozen teacher's forward (mmdet/models/detectors/kd_one_stage.py:70-72: the same
 * ~150 launches on the same buffers every step). */
int ld_record_begin(void);
int64_t ld_record_end(void);
int ld_record_abort(void);
int ld_record_count(int64_t handle);
int ld_record_replay(int64_t handle, ld_stream_t stream);
int ld_record_free(int64_t handle);
/* ---- step lists: a captured hipGraph re-issued as plain stream launches ---------
 * ld_step_list_build walks the nodes of a captured hipGraph_t ONCE (kernel / memcpy
 * / memset parameters, dependency edges, capture order), assigns them to at most
 * max_lanes streams so that the capture's concurrency survives, and returns a
 * handle > 0 (negative: LD_EUNSUPPORTED for host / child-graph / allocation nodes,
 * LD_EINVAL otherwise).  ld_step_list_replay re-issues the step with one C loop of
 * hipLaunchKernel / hipMemcpyAsync / hipMemsetAsync calls (+ an event pair per
 * cross-lane edge): lane 0 is `stream`, the other lanes are streams the list owns;
 * they start behind everything enqueued on `stream` and `stream` ends up behind
 * them.  The graph and every buffer the captured step touched must outlive the
 * list.  ld_step_list_info: counts[8] = {kernel, memcpy, memset, ordering-only
 * nodes, lanes, cross-lane waits, nodes, 0}.  Nothing in the reference (mmcv's
 * runner issues every kernel from Python, mmdet/apis/train.py:74-127): what takes
 * the host language out of the student's ~350 launches per step. */
int64_t ld_step_list_build(void* hip_graph, int max_lanes);
/* Device-to-device copy as ONE kernel launch (16 bytes per lane): what a step that
 * will be captured uses instead of hipMemcpyAsync, whose captured 1-D memcpy node
 * this runtime cannot describe back to ld_step_list_build. */
int ld_copy_d2d(void* dst, const void* src, size_t bytes, ld_stream_t stream);
int ld_step_list_info(int64_t handle, int* counts);
int ld_step_list_replay(int64_t handle, ld_stream_t stream);
int ld_step_list_free(int64_t handle);
/* debugging aid: the node the last failing replay stopped at: out8 = {node index,
 * hipGraphNodeType, hipError_t, lane, four type-specific details} */
int ld_step_list_last_failure(int* out8);
/* Stream `to` continues after everything enqueued on `from` so far (event record
 * + stream wait on a reusable per-device event): torch's
 * `side.wait_stream(main)` as one call.  Nothing in the reference: the fork of a
 * weight gradient onto the background stream of the backward pass.  Not on a
 * capturing stream. */
int ld_stream_fork(ld_stream_t from, ld_stream_t to);

/* ---- one frozen bottleneck as ONE launch (bf16 mode, C8-only activations) -----
 * y = relu(bn3(conv1x1(relu(bn2(conv3x3(relu(bn1(conv1x1(x)))))))) + x) for an
 * identity block (no downsample, stride 1) of a frozen ResNet trunk
 * (mmdet/models/backbones/resnet.py:260-299; the R101 teacher's layer3).  x_c8 /
 * y_c8: (N, Cin/8, H*W, 8) bf16 images (y_c8 != x_c8); w1 / w2 / w3: the bf16
 * forward images of the three convs (ld_conv_bf16_weight_transform: (mid, Cin, 1,
 * 1), (mid, mid, 3, 3), (Cin, mid, 1, 1)); scale / shift: the folded eval-BN
 * coefficients (ld_bn_prepare), 16-byte aligned.  The mid activations stay in
 * LDS; results are bit-identical to the three ld_conv_bf16_forward_c8 launches
 * with C8-only outputs.  ld_bottleneck_c8_supported: which widths the kernel is
 * built for (Cin 1024 / mid 256); LD_EUNSUPPORTED otherwise. */
typedef struct {
  int32_t N, H, W, Cin, mid, reserved;
  const void* w1;
  const void* w2;
  const void* w3;
  const float* scale1;
  const float* shift1;
  const float* scale2;
  const float* shift2;
  const float* scale3;
  const float* shift3;
} ld_bottleneck_t;
int ld_bottleneck_c8_supported(int Cin, int mid, int H, int W);
int ld_bottleneck_c8_forward(const ld_bottleneck_t* b, const void* x_c8, void* y_c8,
                             ld_stream_t stream);
int ld_conv_bf16_dgrad_acc(const ld_conv_t* c, const float* dy, const void* wt_bwd,
                           const float* addend, float* dx, ld_stream_t stream);
int ld_conv_bf16_dgrad_c8_acc(const ld_conv_t* c, const void* dy_c8, const void* wt_bwd,
                              const float* addend, float* dx, ld_stream_t stream);
int ld_conv_bf16_tune_dgrad_c8(const ld_conv_t* c, const void* dy_c8,
                               const void* wt_bwd, float* dx, ld_stream_t stream);
/* Weight gradient with BOTH operands as C8 images (Cin, Cout multiples of 8):
 * tile rows are positions (16-byte loads, aligned under any tap shift), the
 * position -> lane transpose the MFMA needs is done by ds_read_b64_tr_b16 in
 * the LDS read.  fp32 slabs / dw exactly as ld_conv_bf16_wgrad (same workspace
 * size); the per-element summation order differs from it. */
int ld_conv_bf16_wgrad_c8(const ld_conv_t* c, const void* x_c8, const void* dy_c8,
                          float* dw, int accumulate, void* workspace,
                          size_t workspace_bytes, ld_stream_t stream);

/* Small-Cin variant (the 7x7 stride-2 stem, resnet.py:558-570): flat
 * (ci,kh,kw) reduction; wt = [pad32(Cin*KH*KW)][Cout] image obtained with
 * ld_conv_weight_transform(w, Cout, Cin*KH*KW, 1, 1, wt, NULL). */
int ld_conv_forward_smallc(const ld_conv_t* c, const float* x, const float* wt,
                           const ld_conv_epilogue_t* ep, float* y,
                           ld_stream_t stream);

/* ---- normalisation / elementwise layers ----------------------------------
 * Level list of a level-concatenated (N, C, P) tensor (P = sum H_l*W_l). */
typedef struct {
  int32_t num_levels;
  int32_t H[LD_MAX_LEVELS];
  int32_t W[LD_MAX_LEVELS];
} ld_levels_t;

/* BatchNorm2d in eval mode (running stats frozen, affine trainable:
 * resnet.py:639-648, norm_cfg requires_grad=True): scale = gamma*rsqrt(var+eps),
 * shift = beta - mean*scale, rstd = rsqrt(var+eps) (rstd may be NULL). */
int ld_bn_prepare(const float* gamma, const float* beta, const float* mean,
                  const float* var, float eps, int C, float* scale, float* shift,
                  float* rstd, ld_stream_t stream);
/* The same for many BatchNorms in one launch (device-resident job table, as
 * ld_conv_weight_transform_batch). */
typedef struct {
  const float* gamma;
  const float* beta;
  const float* mean;
  const float* var;
  float* scale;
  float* shift;
  float* rstd;
  float eps;
  int32_t C, first_block, reserved;
} ld_bn_job_t;
int ld_bn_prepare_batch(const ld_bn_job_t* jobs, const int32_t* block_job,
                        int nblocks, ld_stream_t stream);
/* y = act(x*scale[c] + shift[c] (+ residual)) on (N, C, P)
 * (Bottleneck/BasicBlock tails, resnet.py:65-92,260-299). */
int ld_bn_act_forward(const float* x, const float* residual, const float* scale,
                      const float* shift, int N, int C, int P, int relu, float* y,
                      ld_stream_t stream);
/* The same with the bf16 channel-blocked image of y as a side output (see
 * ld_gn_forward_c8); C % 8 == 0, P % 4 == 0, 16-byte aligned tensors. */
int ld_bn_act_forward_c8(const float* x, const float* residual, const float* scale,
                         const float* shift, int N, int C, int P, int relu,
                         float* y, void* y_c8, ld_stream_t stream);
size_t ld_bn_act_backward_workspace_bytes(int N, int C, int P);
/* dz = relu ? dy*(y>0) : dy; dx = dz*scale (may be NULL); dres = dz (may be
 * NULL); dgamma = sum dz*(x-mean)*rstd, dbeta = sum dz (either may be NULL). */
int ld_bn_act_backward(const float* dy, const float* y, const float* x,
                       const float* scale, const float* mean, const float* rstd,
                       int N, int C, int P, int relu, float* dx, float* dres,
                       float* dgamma, float* dbeta, int accumulate,
                       void* workspace, size_t workspace_bytes,
                       ld_stream_t stream);
/* The same with the bf16 channel-blocked image of dx as a side output (bf16
 * mode: dx feeds the conv's C8 data- / weight-gradient kernels directly, no
 * conversion launch).  C % 8 == 0, P % 4 == 0, 16-byte aligned tensors,
 * N * ceil(P / 256) <= 256 partial slots; LD_EUNSUPPORTED otherwise (use the
 * plain form + ld_conv_to_c8). */
int ld_bn_act_backward_c8(const float* dy, const float* y, const float* x,
                          const float* scale, const float* mean, const float* rstd,
                          int N, int C, int P, int relu, float* dx, void* dx_c8,
                          float* dres, float* dgamma, float* dbeta, int accumulate,
                          void* workspace, size_t workspace_bytes, ld_stream_t stream);
/* ld_bn_act_backward_c8 with the two saved activations given as bf16 C8 images
 * (N, C/8, P, 8) instead of fp32 tensors (round 6): y_c8 = the image of the
 * layer's output (only its sign is used, the ReLU mask; NULL without relu), x_c8 =
 * the conv result before the affine as written through
 * ld_conv_epilogue_t.y_raw_c8.  dx may be NULL (only dx_c8 is written).  8 instead
 * of 12 bytes read per element; d(gamma) sees the bf16-rounded conv result.  Needs
 * C % 8 == 0 and P % 2 == 0 (four positions per thread when P % 4 == 0, two
 * otherwise: the 25 x 42 stage). */
int ld_bn_act_backward_c8in(const float* dy, const void* y_c8, const void* x_c8,
                            const float* scale, const float* mean, const float* rstd,
                            int N, int C, int P, int relu, float* dx, void* dx_c8,
                            float* dres, float* dgamma, float* dbeta, int accumulate,
                            void* workspace, size_t workspace_bytes,
                            ld_stream_t stream);
/* Deferred finalisation (round 5, see ld_wgrad_reduce_batch): with accumulate ==
 * LD_GRAD_DEFER the two BN backward entry points above write only their
 * per-workgroup fp64 partials ([C][nsplit] pairs (sum dz, sum dz*xhat), nsplit =
 * ld_bn_act_backward_nsplit(N, C, P, c8), c8 = 0 for ld_bn_act_backward, 1 for
 * ld_bn_act_backward_c8, 2 for ld_bn_act_backward_c8in) into `workspace` -- which the caller
 * then owns until ld_bn_bwd_finalize_batch has summed every job of a bucket in ONE
 * launch (16 channels per block, a job owns ceil(C / 16) blocks from first_block;
 * the arithmetic of the per-layer finalize launch: 37 launches per step fewer).
 * dgamma / dbeta must be non-NULL in the deferring call (they request the
 * partials) and are not written by it. */
#define LD_GRAD_DEFER 2
typedef struct {
  const double* partial;
  float* dgamma;
  float* dbeta;
  int32_t C, nsplit, accumulate, first_block;
} ld_bn_fin_job_t;
int ld_bn_act_backward_nsplit(int N, int C, int P, int c8);
int ld_bn_bwd_finalize_batch(const ld_bn_fin_job_t* jobs, const int32_t* block_job,
                             int nblocks, ld_stream_t stream);
/* db[c] = sum_{n,p} dy (conv bias gradient, fpn.py / gfl_cls / gfl_reg). */
int ld_bias_grad(const float* dy, int N, int C, int P, float* db, int accumulate,
                 ld_stream_t stream);
/* The same sum as [C][nsplit] fp64 partial pairs (sum, 0), nsplit =
 * ld_bias_grad_nsplit(N, C, P), for ld_bn_bwd_finalize_batch (a job with dgamma ==
 * NULL, dbeta = db): the bias gradients of a bucket are finalised with its norm
 * gradients in one launch. */
int ld_bias_grad_nsplit(int N, int C, int P);
int ld_bias_grad_partial(const float* dy, int N, int C, int P, void* partial,
                         size_t partial_bytes, ld_stream_t stream);
/* GroupNorm(G) (+ReLU) applied per FPN level of a level-concatenated tensor
 * (gfl_head.py:102-126: each level is normalised on its own).  mean/rstd:
 * (N, G, num_levels) outputs kept for the backward. */
size_t ld_gn_forward_workspace_bytes(const ld_levels_t* lv, int N, int G);
int ld_gn_forward(const ld_levels_t* lv, const float* x, const float* gamma,
                  const float* beta, int N, int C, int G, float eps, int relu,
                  float* y, float* mean, float* rstd, void* workspace,
                  size_t workspace_bytes, ld_stream_t stream);
size_t ld_gn_backward_workspace_bytes(const ld_levels_t* lv, int N, int C);
int ld_gn_backward(const ld_levels_t* lv, const float* dy, const float* y,
                   const float* x, const float* gamma, const float* mean,
                   const float* rstd, int N, int C, int G, int relu, float* dx,
                   float* dgamma, float* dbeta, int accumulate, void* workspace,
                   size_t workspace_bytes, ld_stream_t stream);
/* The same with a bf16 channel-blocked side output (see ld_conv_to_c8): y_c8 /
 * dx_c8 receive the (N, C/8, P, 8) bf16 image of the fp32 tensor written in the
 * same launch, for the bf16 conv that consumes it next.  The fp32 results are
 * bit-identical to ld_gn_forward / ld_gn_backward.  Needs C % 8 == 0, P % 4 == 0
 * and 16-byte aligned tensors (LD_EUNSUPPORTED otherwise). */
int ld_gn_forward_c8(const ld_levels_t* lv, const float* x, const float* gamma,
                     const float* beta, int N, int C, int G, float eps, int relu,
                     float* y, void* y_c8, float* mean, float* rstd, void* workspace,
                     size_t workspace_bytes, ld_stream_t stream);
int ld_gn_backward_c8(const ld_levels_t* lv, const float* dy, const float* y,
                      const float* x, const float* gamma, const float* mean,
                      const float* rstd, int N, int C, int G, int relu, float* dx,
                      void* dx_c8, float* dgamma, float* dbeta, int accumulate,
                      void* workspace, size_t workspace_bytes, ld_stream_t stream);
/* The lean form of ld_gn_backward_c8 (round 6, bf16 mode): y is not an operand --
 * the ReLU mask y > 0 is recomputed from x, mean, rstd, gamma, beta with the
 * forward's own expression (bit-identical to reading y back) -- and dx (fp32) may
 * be NULL when every consumer of the gradient takes the C8 image dx_c8.  18 instead
 * of 30 bytes of HBM traffic per element.  ld_gn_forward_c8 accepts y == NULL the
 * same way (only the C8 image of the output is written). */
int ld_gn_backward_c8_lean(const ld_levels_t* lv, const float* dy, const float* x,
                           const float* gamma, const float* beta, const float* mean,
                           const float* rstd, int N, int C, int G, int relu, float* dx,
                           void* dx_c8, float* dgamma, float* dbeta, int accumulate,
                           void* workspace, size_t workspace_bytes, ld_stream_t stream);
/* MaxPool2d(kernel 3, stride 2, pad 1) on rows = N*C planes (resnet.py:570);
 * forward only (the stem is frozen, frozen_stages=1). */
int ld_maxpool3x3s2(const float* x, int rows, int H, int W, float* y,
                    ld_stream_t stream);
/* FPN top-down step (fpn.py:182-191): out = fine + nearest_up(coarse) with the
 * finer map's size as target; rows = N*C planes.  Backward: dfine = dout
 * (identity), dcoarse[q] = sum of dout over the fine cells that read q. */
int ld_upsample_add_forward(const float* fine, const float* coarse, int rows,
                            int Hf, int Wf, int Hc, int Wc, float* out,
                            ld_stream_t stream);
int ld_upsample_add_backward(const float* dout, int rows, int Hf, int Wf, int Hc,
                             int Wc, float* dcoarse, ld_stream_t stream);
/* ... + addend (the gradient the coarse map already holds from its other
 * consumer, the level's output conv; may be NULL; may be dcoarse itself). */
int ld_upsample_add_backward_acc(const float* dout, int rows, int Hf, int Wf, int Hc,
                                 int Wc, const float* addend, float* dcoarse,
                                 ld_stream_t stream);
/* Level packing for the shared head towers (gfl_head.py:164-172 applies the same
 * convs to every FPN level; here they run on ONE level-concatenated (N, C, P)
 * tensor): levels[l] = contiguous (rows, H_l*W_l) fp32, x3 = (rows, P), rows = N*C.
 * pack copies the levels into x3, unpack is its inverse (the backward): one
 * launch for all levels. */
int ld_pack_levels(const ld_levels_t* lv, const float* const* levels, int rows,
                   float* x3, ld_stream_t stream);
int ld_unpack_levels(const ld_levels_t* lv, const float* x3, int rows,
                     float* const* levels, ld_stream_t stream);
/* The same with the bf16 C8 image ((N, C/8, len, 8), see ld_conv_to_c8) of the
 * destination as a side output: x3_c8 for pack, levels_c8[l] for unpack; C % 8 == 0. */
int ld_pack_levels_c8(const ld_levels_t* lv, const float* const* levels, int N, int C,
                      float* x3, void* x3_c8, ld_stream_t stream);
int ld_unpack_levels_c8(const ld_levels_t* lv, const float* x3, int N, int C,
                        float* const* levels, void* const* levels_c8, ld_stream_t stream);
/* mmcv Scale per level (gfl_head.py:182): y = x * scales[level]. */
int ld_scale_levels_forward(const ld_levels_t* lv, const float* x,
                            const float* scales, int rows, float* y,
                            ld_stream_t stream);
size_t ld_scale_levels_backward_workspace_bytes(const ld_levels_t* lv);
int ld_scale_levels_backward(const ld_levels_t* lv, const float* dy, const float* x,
                             const float* scales, int rows, float* dx,
                             float* dscales, int accumulate, void* workspace,
                             size_t workspace_bytes, ld_stream_t stream);
/* torch.optim.SGD(momentum, weight_decay) over a flat parameter arena
 * (apis/train.py:88): d = g*grad_scale + wd*p; buf = mu*buf + d; p -= lr*buf. */
int ld_sgd_step(float* params, const float* grads, float* momentum_buf, size_t n,
                float lr, float momentum, float weight_decay, float grad_scale,
                ld_stream_t stream);
/* The same update with hyper = {lr, momentum, weight_decay, grad_scale} read
 * from DEVICE memory (bit-identical arithmetic): inside a captured hipGraph the
 * by-value arguments above are frozen, a buffer is not -- the host rewrites it
 * between replays (learning-rate schedules, apis/train.py:88 + lr hooks). */
int ld_sgd_step_dev(float* params, const float* grads, float* momentum_buf, size_t n,
                    const float* hyper, ld_stream_t stream);

/* ---- GFLv2 distribution-guided quality branch (config 5, R-V2) ----------------
 * GFocalHead.forward_single's tail (gfocal_head.py:201-217) for all levels and
 * images in one launch, on the level-concatenated (N, C, P) tensors:
 *   prob = softmax over the 17 bins of each side of reg (N, 68, P)
 *   stat = [top-4 probabilities (descending), their mean] per side -> 20 values
 *   quality = sigmoid(w2 . relu(w1 stat + b1) + b2)       reg_conf, :140-144
 *   cls_score[c] = sigmoid(cls_feat[c]) * quality          (N, C, P)
 * w1 (64, 20) row-major, b1 (64), w2 (64), b2 (1); reg_topk = 4, add_mean,
 * reg_channels = 64 are compiled in.  quality (N, P) is kept for the backward.
 * Backward: g_cls_feat = g_cls_score * quality * sigma'(cls_feat);
 * g_reg (N, 68, P) through sigmoid, the MLP, top-k scatter and the softmax
 * (both fully overwritten); parameter gradients summed over all anchors in a
 * fixed order (per-block partials in the workspace), written or accumulated. */
int ld_quality_forward(const float* reg, const float* cls_feat, int N, int C, int P,
                       const float* w1, const float* b1, const float* w2,
                       const float* b2, float* cls_score, float* quality,
                       ld_stream_t stream);
size_t ld_quality_backward_workspace_bytes(int N, int P);
int ld_quality_backward(const float* reg, const float* cls_feat, const float* quality,
                        const float* g_cls_score, int N, int C, int P,
                        const float* w1, const float* b1, const float* w2,
                        const float* b2, float* g_cls_feat, float* g_reg,
                        float* g_w1, float* g_b1, float* g_w2, float* g_b2,
                        int accumulate, void* workspace, size_t workspace_bytes,
                        ld_stream_t stream);

/* ---- measurement support: HBM streaming ceilings ------------------------------
 * Not on the train step.  bench.py times these in the same process as the
 * LD-KL kernel so its roofline fraction can be read against what the memory
 * system delivers for (a) a plain copy of n floats (width = 1 or 4 floats per
 * lane, nt = non-temporal access; 8 n bytes move) and (b) the LD-KL kernel's own
 * access pattern without its arithmetic: s, t, g are (68, rows) channel-major
 * maps, every (row, side) thread reads 17 + 17 planes and writes 17 (204 bytes
 * per row-side); side_fast = the four sides of a 256-row chunk in adjacent
 * workgroups (the mapping the train step's kernel uses). */
int ld_probe_copy(const float* src, float* dst, int64_t n, int width, int nt,
                  ld_stream_t stream);
int ld_probe_planes(const float* s, const float* t, float* g, int64_t rows, int nt,
                    int side_fast, ld_stream_t stream);

/* ---- grouped convolution, forward only (config 5's X-101 teacher) --------------
 * The 3x3 conv2 of ResNeXt's Bottleneck (mmdet/models/backbones/resnext.py:49-61,
 * nn.Conv2d(width, width, 3, stride, padding=1, groups=32, bias=False)) and,
 * with K = 1, the grouped GEMM behind a grouped deformable conv (:62-74 over
 * ld_deform_im2col's columns).  x (N, Cin, Hin*Win), y (N, Cout, Hout*Wout)
 * fp32 NCHW; Cin % groups == Cout % groups == 0, Cout / groups in {4, 8, 16, 32},
 * K in {1, 3}.  wimage = ld_gconv_weight_transform of the PyTorch weight
 * (Cout, Cin/groups, K, K): [group][ci][tap][co], so the Cout/groups weights of
 * one (ci, tap) are consecutive dwords at a wave-uniform address (scalar loads).
 * Epilogue: y = relu?(scale[c] * acc + shift[c]) (eval-mode BN folded in; both
 * NULL = plain conv).  VALU kernel: 9-72 flop/B, HBM/L1-bound, not MFMA work. */
size_t ld_gconv_weight_image_floats(int Cout, int Cin, int groups, int K);
int ld_gconv_weight_transform(const float* w, int Cout, int Cin, int groups, int K,
                              float* image, ld_stream_t stream);
int ld_gconv_forward(const float* x, const float* wimage, float* y, int N, int Cin,
                     int Cout, int groups, int K, int stride, int pad, int Hin, int Win,
                     const float* scale, const float* shift, int relu,
                     ld_stream_t stream);

/* ---- deformable convolution v1, forward only (config 4's R101-DCN teacher) ---
 * mmcv.ops.DeformConv2dPack under resnet.py:171-194 (deform_groups = 1,
 * groups = 1).  ld_deform_im2col samples x (N, Cin, Hin, Win) bilinearly at
 * p*stride - pad + k*dilation + offset and writes the column tensor
 * col (N, Cin*KH*KW, Hout*Wout), channel = ci*KH*KW + k (= the order of
 * weight.view(Cout, Cin*KH*KW)); offset (N, 2*KH*KW, Hout, Wout) holds (dy, dx)
 * per tap, from the layer's own conv_offset 3x3 conv (ld_conv_forward).  The
 * product with the weights is ld_conv_forward as a 1x1 conv over Cin*KH*KW
 * channels.  Outside (-1, H) x (-1, W) the sample is 0; neighbours outside the
 * map contribute 0 (mmcv deform_conv_cuda_kernel.cuh; parity unpinned, the op
 * is not in the reference checkout). */
int ld_deform_im2col(const float* x, const float* offset, int N, int Cin, int Hin,
                     int Win, int KH, int KW, int stride, int pad, int dilation,
                     float* col, ld_stream_t stream);

/* ---- device input pipeline ---------------------
 * Resize(keep_ratio) -> flip -> Normalize(to_rgb) -> Pad -> collate of the
 * reference's train_pipeline (configs/ld/ld_r18_gflv1_r101_fpn_coco_1x.py:66-77,
 * datasets/pipelines/transforms.py:203-233,416-450,524-580) for a whole batch in
 * one launch.  `imgs` is a DEVICE array of N descriptors; `data` points to a
 * decoded uint8 HWC image (3 channels, BGR as cv2.imread delivers) in device
 * memory; (new_h, new_w) is the resized size (mmcv.imrescale's rounding is host
 * logic: ld_amd/pipeline.py); `flip` = horizontal flip of the resized image.
 * out (N, 3, Hpad, Wpad) fp32, zeros beyond (new_h, new_w); mean / std_inv are
 * HOST float[3] in output-channel order (RGB when to_rgb).  Bilinear =
 * cv2 INTER_LINEAR on 8-bit data (11-bit fixed-point coefficients, rounded
 * uint8 result before normalisation). */
typedef struct {
  const unsigned char* data;
  int32_t src_h, src_w, new_h, new_w, flip, reserved;
} ld_image_t;
int ld_preprocess_batch(const ld_image_t* imgs, int N, int Hpad, int Wpad,
                        const float* mean, const float* std_inv, int to_rgb,
                        float* out, ld_stream_t stream);

/* ---- inference post-processing ---------------
 * GFLHead.get_bboxes for a whole batch (gfl_head.py:354-451 ->
 * post_processing/bbox_nms.py:70-195 -> mmcv.ops.batched_nms): sigmoid scores,
 * Integral * stride, per-level top-nms_pre by max class score, distance2bbox +
 * clamp to img_hw, optional division by scale_factors (rescale=True), score
 * threshold, class-aware greedy NMS (IoU > iou_thr suppresses), best
 * max_per_img (<= 1024) detections per image in descending score order.
 *   cls / reg      NCHW-direct head maps, (N, num_classes, H_l, W_l) /
 *                  (N, 4*(reg_max+1), H_l, W_l); reg_max must be 16
 *   img_hw         device (N, 2): img_shape height, width
 *   scale_factors  device (N, 4) or NULL
 *   dets           device (N, max_per_img, 5): x1, y1, x2, y2, score
 *   labels         device (N, max_per_img) int64;  counts device (N) int32
 * Equal scores: lower (anchor, class) index first.  iou_thr must be >= 0
 * (LD_EINVAL otherwise: a negative threshold suppresses across classes in the
 * reference's class-shift formulation).  Synchronises the stream once. */
size_t ld_get_bboxes_workspace_bytes(const ld_geom_t* g, int num_classes,
                                     int nms_pre);
int ld_get_bboxes(const ld_geom_t* g, const ld_maps_t* cls, const ld_maps_t* reg,
                  int num_classes, int reg_max, const float* img_hw,
                  const float* scale_factors, int nms_pre, float score_thr,
                  float iou_thr, int max_per_img, float* dets, int64_t* labels,
                  int32_t* counts, void* workspace, size_t workspace_bytes,
                  ld_stream_t stream);
/* The reference's optional nms type 'voting_cluster_diounms'
 * (post_processing/bbox_nms.py:141-176): same pipeline, but (1) the overlap
 * measure of the suppression is DIoU with the centre-distance term raised to
 * 0.8, evaluated on boxes shifted by 4000 * label as the reference shifts them
 * (Cluster-NMS iterated to its fixed point keeps exactly the boxes of the greedy
 * pass used here), and (2) score voting: every kept box is replaced by the
 * exp(-(1 - DIoU)^2 / 0.025) * score weighted mean of the candidates at or
 * below it in score order whose DIoU with it exceeds 0.7 (all other candidates
 * enter with the factor exp(-40), as in the reference's dense product).
 * Same arguments, workspace and outputs as ld_get_bboxes. */
int ld_get_bboxes_voting(const ld_geom_t* g, const ld_maps_t* cls, const ld_maps_t* reg,
                  int num_classes, int reg_max, const float* img_hw,
                  const float* scale_factors, int nms_pre, float score_thr,
                  float iou_thr, int max_per_img, float* dets, int64_t* labels,
                  int32_t* counts, void* workspace, size_t workspace_bytes,
                  ld_stream_t stream);
/* Both variants behind one entry point.  flags: LD_INFER_VOTING = the
 * score-voting Cluster-DIoU-NMS above; LD_INFER_PROB = the class maps already
 * hold probabilities and no sigmoid is applied -- GFocalHead.get_bboxes
 * (gfocal_head.py:317-596: cls_score = sigmoid(cls) * quality comes out of the
 * head, and num_classes = 81 there because use_sigmoid=False makes the
 * background column an ordinary score channel, anchor_head.py:68-71).
 * ``ctr`` (nullable): per-level (N, 1, H, W) centerness logits of ATSSGFLHead /
 * FCOSGFLHead (atss_gfl_head.py:420-575, fcos_gfl_head.py:347-546): the top-k
 * key becomes max_c score_c * sigmoid(centerness) and the factor multiplies
 * every candidate's score AFTER the score_thr test (multiclass_nms
 * score_factors, bbox_nms.py:114-123); not combinable with LD_INFER_VOTING.
 * LD_INFER_POINTS: decode about the FCOS points (x, y) * stride + stride / 2
 * (fcos_gfl_head.py:548-558) instead of the anchor centres (x, y) * stride.
 * ``num_base`` > 1 (RetinaGFLHead._get_bboxes, retina_gfl_head.py:301-412): the
 * maps are (N, num_base * C, H, W) / (N, num_base * 68, H, W); candidates are
 * the rows (cell, base anchor) in that order, top-k per level over all of
 * them, decoded about the shared cell centre.  Workspace: the _ex_ size. */
#define LD_INFER_VOTING 1
#define LD_INFER_PROB 2
#define LD_INFER_POINTS 4
size_t ld_get_bboxes_ex_workspace_bytes(const ld_geom_t* g, int num_classes,
                                       int num_base, int nms_pre);
int ld_get_bboxes_ex(const ld_geom_t* g, const ld_maps_t* cls, const ld_maps_t* reg,
                     const ld_maps_t* ctr, int num_classes, int num_base,
                     int reg_max, const float* img_hw,
                     const float* scale_factors, int nms_pre, float score_thr,
                     float iou_thr, int max_per_img, int flags, float* dets,
                     int64_t* labels, int32_t* counts, void* workspace,
                     size_t workspace_bytes, ld_stream_t stream);
/* get_bboxes(with_nms=False) of the same heads: stages 1-3 only (per-level top
 * nms_pre, Integral * stride, decode + clamp, optional rescale).  Outputs, per
 * image, the K = ld_get_bboxes_num_selected() rows in level order (within a
 * sorted level: descending key, as torch.topk returns them): boxes (N, K, 4),
 * scores (N, K, num_classes) WITHOUT the reference's zero background column,
 * and, with ``ctr``, the centerness factors (N, K).  Workspace: the _ex_ size. */
int ld_get_bboxes_num_selected(const ld_geom_t* g, int num_base, int nms_pre);
int ld_get_bboxes_pre_nms(const ld_geom_t* g, const ld_maps_t* cls,
                          const ld_maps_t* reg, const ld_maps_t* ctr,
                          int num_classes, int num_base, int reg_max,
                          const float* img_hw, const float* scale_factors,
                          int nms_pre, int flags, float* boxes, float* scores,
                          float* factors, void* workspace, size_t workspace_bytes,
                          ld_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* LD_HIP_H_ */
