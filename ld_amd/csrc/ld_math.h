// Scalar math shared by the LD loss / target kernels.
//
// Every function is `LD_HD` (host + device) and free of wave intrinsics so the
// exact same arithmetic can be compiled by g++ into the host-side unit-test
// harness (tests/host_harness.cpp, CPU tests only) and by hipcc into the
// gfx950 kernels (the product).  Reference formulas are cited per function
// (paths relative to the reference checkout).
//
// This translation unit family is compiled with -ffp-contract=off: target
// assignment must reproduce the reference's fp32 op sequence bit-for-bit
// (SURVEY.md K12), which an FMA contraction would break.
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define LD_HD __host__ __device__ __forceinline__
#else
#define LD_HD inline
#endif

namespace ld {

constexpr int kRegBins = 17;   // reg_max + 1 (gfl_head.py:86 reg_max=16)
constexpr float kNegInf = -100000000.0f;  // atss_assigner.py:63 INF

struct Box {
  float x1, y1, x2, y2;
};

LD_HD float fmaxf_(float a, float b) { return a > b ? a : b; }
LD_HD float fminf_(float a, float b) { return a < b ? a : b; }

// Anchor of level stride `s` at grid cell (x, y): square of side 8*s centred on
// (x*s, y*s) -- anchor_generator.py:142-185 (center_offset = 0,
// octave_base_scale = 8, ratio 1) and :229-270 (shift + base anchor).
LD_HD Box anchor_box(int x, int y, int stride, float half_side) {
  float cx = (float)(x * stride), cy = (float)(y * stride);
  return Box{cx - half_side, cy - half_side, cx + half_side, cy + half_side};
}

// iou2d_calculator.py:111-171, mode 'iou' (pairwise), eps = 1e-6.
LD_HD float iou_pair(const Box& a, const Box& b) {
  float area1 = (a.x2 - a.x1) * (a.y2 - a.y1);
  float area2 = (b.x2 - b.x1) * (b.y2 - b.y1);
  float w = fmaxf_(fminf_(a.x2, b.x2) - fmaxf_(a.x1, b.x1), 0.0f);
  float h = fmaxf_(fminf_(a.y2, b.y2) - fmaxf_(a.y1, b.y1), 0.0f);
  float overlap = w * h;
  float uni = area1 + area2 - overlap;
  uni = fmaxf_(uni, 1e-6f);
  return overlap / uni;
}

// iou2d_calculator.py:123-126,179-188, mode 'diou' -- fork-specific and
// IoF-based (quirk Q1): inter/area(a) - rho^2/c^2.
LD_HD float diou_pair(const Box& a, const Box& b) {
  float area1 = (a.x2 - a.x1) * (a.y2 - a.y1);
  float w = fmaxf_(fminf_(a.x2, b.x2) - fmaxf_(a.x1, b.x1), 0.0f);
  float h = fmaxf_(fminf_(a.y2, b.y2) - fmaxf_(a.y1, b.y1), 0.0f);
  float overlap = w * h;
  float uni = fmaxf_(area1, 1e-6f);
  float ious = overlap / uni;
  float l = (b.x1 + b.x2) - (a.x1 + a.x2);
  float r = (b.y1 + b.y2) - (a.y1 + a.y2);
  float left = (l * l) / 4.0f;
  float right = (r * r) / 4.0f;
  float rho2 = left + right;
  float ew = fmaxf_(fmaxf_(a.x2, b.x2) - fminf_(a.x1, b.x1), 0.0f);
  float eh = fmaxf_(fmaxf_(a.y2, b.y2) - fminf_(a.y1, b.y1), 0.0f);
  float ec = ew * ew + eh * eh;
  ec = fmaxf_(ec, 1e-6f);
  return ious - rho2 / ec;
}

// Centre distance, atss_assigner.py:93-103: (d).pow(2).sum(-1).sqrt()
LD_HD float centre_dist(float ax, float ay, float gx, float gy) {
  float dx = ax - gx, dy = ay - gy;
  float s = dx * dx + dy * dy;
  return sqrtf(s);
}

LD_HD float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// BCE-with-logits(x, 0) = softplus(x), computed as torch does:
// max(x,0) + log1p(exp(-|x|))
LD_HD float softplusf_(float x) {
  return fmaxf_(x, 0.0f) + log1pf(expf(-fabsf(x)));
}

// Quality focal loss element, beta = 2 (gfocal_loss.py:8-50).
//   negative entry : q = softplus(x) * sigma^2
//   positive entry : q = (softplus(x) - s*x) * (s - sigma)^2     (s = score)
// Returns q, writes dq/dx.
LD_HD float qfl_neg(float x, float* dq) {
  float s = sigmoidf_(x), sp = softplusf_(x);
  *dq = s * s * s + 2.0f * s * s * (1.0f - s) * sp;
  return sp * s * s;
}
LD_HD float qfl_pos(float x, float score, float* dq) {
  float s = sigmoidf_(x), sp = softplusf_(x);
  float bce = sp - score * x;
  float d = score - s;
  *dq = (s - score) * d * d - 2.0f * d * s * (1.0f - s) * bce;
  return bce * d * d;
}

// Sigmoid focal loss, gamma = 2 (losses/focal_loss.py:12-47, the pure-torch
// py_sigmoid_focal_loss the reference runs on CPU; the compiled mmcv op computes
// the same function):  bce_with_logits(x, t) * (alpha t + (1 - alpha)(1 - t)) *
// pt^2,  pt = (1 - p) t + p (1 - t).
//   t = 1 : f = alpha (1 - p)^2 softplus(-x),  df/dx = alpha (1 - p)^2 (2 p log p - (1 - p))
//   t = 0 : f = (1 - alpha) p^2 softplus(x),   df/dx = (1 - alpha) p^2 (p - 2 (1 - p) log(1 - p))
// (-log p = softplus(-x), -log(1 - p) = softplus(x))
LD_HD float focal_pos(float x, float alpha, float* df) {
  float p = sigmoidf_(x), spn = softplusf_(-x);
  float q = 1.0f - p;
  *df = alpha * q * q * (-2.0f * p * spn - q);
  return alpha * q * q * spn;
}
LD_HD float focal_neg(float x, float alpha, float* df) {
  float p = sigmoidf_(x), sp = softplusf_(x);
  *df = (1.0f - alpha) * p * p * (p + 2.0f * (1.0f - p) * sp);
  return (1.0f - alpha) * p * p * sp;
}
// binary_cross_entropy_with_logits(x, t) and d/dx (centerness,
// losses/cross_entropy_loss.py:52-90)
LD_HD float bce_logits(float x, float t, float* d) {
  *d = sigmoidf_(x) - t;
  return softplusf_(x) - t * x;
}

// Quality focal loss on PROBABILITIES (use_sigmoid=False: GFLv2, where the
// prediction is sigmoid(cls_feat) * quality; gfocal_loss.py:27-46 with
// F.binary_cross_entropy).  torch clamps the two logs at -100 in the forward
// and differentiates BCE as (p - t) / max(p (1 - p), 1e-12) in the backward
// (aten binary_cross_entropy / _backward), which is reproduced here.
//   negative entry : q = bce(p, 0) * p^2
//   positive entry : q = bce(p, s) * |s - p|^2
LD_HD float bce_prob(float p, float t, float* dbce) {
  float lp = fmaxf_(logf(p), -100.0f), l1p = fmaxf_(logf(1.0f - p), -100.0f);
  *dbce = (p - t) / fmaxf_((1.0f - p) * p, 1e-12f);
  return -(t * lp + (1.0f - t) * l1p);
}
LD_HD float qfl_prob_neg(float p, float* dq) {
  float db;
  float bce = bce_prob(p, 0.0f, &db);
  *dq = db * p * p + bce * 2.0f * p;
  return bce * p * p;
}
LD_HD float qfl_prob_pos(float p, float score, float* dq) {
  float db;
  float bce = bce_prob(p, score, &db);
  float d = score - p;
  *dq = db * d * d - 2.0f * d * bce;
  return bce * d * d;
}

// GIoU loss 1 - GIoU(pred, target) on aligned boxes (iou_loss.py:85-102,
// iou2d_calculator.py:117-177) and d(loss)/d(pred box), taking the same
// sub-gradients autograd takes through clamp(min=0) / max(., eps) / min / max
// (ties in min/max split 0.5/0.5 like torch.maximum's backward).
// Also returns the plain IoU (the QFL quality score, ld_head.py:204-207).
LD_HD float sel_gt(float a, float b) { return a > b ? 1.0f : (a == b ? 0.5f : 0.0f); }
LD_HD float sel_lt(float a, float b) { return a < b ? 1.0f : (a == b ? 0.5f : 0.0f); }

LD_HD float giou_loss_grad(const Box& p, const Box& t, float eps, float* iou_out,
                           float g[4]) {
  float pw = p.x2 - p.x1, ph = p.y2 - p.y1;
  float a1 = pw * ph;
  float a2 = (t.x2 - t.x1) * (t.y2 - t.y1);
  float iw_raw = fminf_(p.x2, t.x2) - fmaxf_(p.x1, t.x1);
  float ih_raw = fminf_(p.y2, t.y2) - fmaxf_(p.y1, t.y1);
  float iw = fmaxf_(iw_raw, 0.0f), ih = fmaxf_(ih_raw, 0.0f);
  float inter = iw * ih;
  float union_raw = a1 + a2 - inter;
  float uni = fmaxf_(union_raw, eps);
  float ew_raw = fmaxf_(p.x2, t.x2) - fminf_(p.x1, t.x1);
  float eh_raw = fmaxf_(p.y2, t.y2) - fminf_(p.y1, t.y1);
  float ew = fmaxf_(ew_raw, 0.0f), eh = fmaxf_(eh_raw, 0.0f);
  float earea_raw = ew * eh;
  float earea = fmaxf_(earea_raw, eps);
  float iou = inter / uni;
  float giou = iou - (earea - uni) / earea;
  *iou_out = iou;
  // giou = I/U - 1 + U/E
  float dg_dI = 1.0f / uni;
  float dg_dU = -inter / (uni * uni) + 1.0f / earea;
  float dg_dE = -uni / (earea * earea);
  float u_live = union_raw > eps ? 1.0f : 0.0f;
  float e_live = earea_raw > eps ? 1.0f : 0.0f;
  float gI = dg_dI - dg_dU * u_live;
  float gA1 = dg_dU * u_live;
  float gE = dg_dE * e_live;
  float g_iw = gI * ih * (iw_raw >= 0.0f ? 1.0f : 0.0f);
  float g_ih = gI * iw * (ih_raw >= 0.0f ? 1.0f : 0.0f);
  float g_ew = gE * eh * (ew_raw >= 0.0f ? 1.0f : 0.0f);
  float g_eh = gE * ew * (eh_raw >= 0.0f ? 1.0f : 0.0f);
  // d giou / d p, negated for the loss
  g[0] = -(gA1 * (-ph) - g_iw * sel_gt(p.x1, t.x1) - g_ew * sel_lt(p.x1, t.x1));
  g[1] = -(gA1 * (-pw) - g_ih * sel_gt(p.y1, t.y1) - g_eh * sel_lt(p.y1, t.y1));
  g[2] = -(gA1 * ph + g_iw * sel_lt(p.x2, t.x2) + g_ew * sel_gt(p.x2, t.x2));
  g[3] = -(gA1 * pw + g_ih * sel_lt(p.y2, t.y2) + g_eh * sel_gt(p.y2, t.y2));
  return 1.0f - giou;
}

// One 17-bin side of the LD regression distribution.
// Inputs: student logits s[17], teacher logits t[17].
//   kl    = T^2/K * sum_k pt_k (log pt_k - log ps_k),  p = softmax(./T)
//           (kd_loss.py:27-36; K = 17 because of .mean(1))
//   dkl_k = T/K * (ps_k - pt_k)
// Outputs are accumulated by the caller with its own weights.
struct SideKL {
  float kl;
  float lse_s, lse_t;  // log-sum-exp of s/T and t/T (max-shifted form folded in)
};

template <int K>
LD_HD float kl_rows(const float* s, const float* t, float invT, float T,
                    float* ps_minus_pt /* [K] out: ps_k - pt_k */) {
  float ms = s[0], mt = t[0];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    ms = fmaxf_(ms, s[k]);
    mt = fmaxf_(mt, t[k]);
  }
  float es[K], et[K];
  float zs = 0.0f, zt = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    es[k] = expf((s[k] - ms) * invT);
    et[k] = expf((t[k] - mt) * invT);
    zs += es[k];
    zt += et[k];
  }
  float rzs = 1.0f / zs, rzt = 1.0f / zt;
  float lzs = logf(zs), lzt = logf(zt);
  float acc = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float pt = et[k] * rzt;
    float ps = es[k] * rzs;
    // log pt - log ps
    float d = ((t[k] - mt) * invT - lzt) - ((s[k] - ms) * invT - lzs);
    acc += pt * d;
    ps_minus_pt[k] = ps - pt;
  }
  return acc * (T * T) / (float)K;
}

// Register-lean form of kl_rows for the dense sweep (round 3): the same KL
// (kd_loss.py:27-36) with the row statistics folded so that nothing but the two
// logit rows stays live --
//   kl = T^2/K * [ 1/(T z_t) * sum_k e_t,k ((t_k - m_t) - (s_k - m_s)) + log z_s - log z_t ]
// (sum_k pt_k = 1 moves the two log-sum-exp terms out of the sum).  On return
// s[k] = ps_k - pt_k (the gradient direction) and t[k] = e_t,k; 34 live values
// instead of kl_rows' 85.  FAST: exp / reciprocal through the hardware
// v_exp_f32 / v_rcp_f32 (1 ulp) instead of the correctly rounded library forms.
#if defined(__HIP_DEVICE_COMPILE__)
LD_HD float fast_exp2_(float x) { return __builtin_amdgcn_exp2f(x); }
LD_HD float fast_rcp_(float x) { return __builtin_amdgcn_rcpf(x); }
#else
LD_HD float fast_exp2_(float x) { return exp2f(x); }
LD_HD float fast_rcp_(float x) { return 1.0f / x; }
#endif

template <int K, bool FAST>
LD_HD float kl_rows_inplace(float* s, float* t, float invT, float T) {
  float ms = s[0], mt = t[0];
#pragma unroll
  for (int k = 1; k < K; ++k) {
    ms = fmaxf_(ms, s[k]);
    mt = fmaxf_(mt, t[k]);
  }
  const float c = FAST ? invT * 1.44269504088896341f : invT;
  float zs = 0.0f, zt = 0.0f, acc = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    const float ds = s[k] - ms, dt = t[k] - mt;
    const float es = FAST ? fast_exp2_(ds * c) : expf(ds * c);
    const float et = FAST ? fast_exp2_(dt * c) : expf(dt * c);
    zs += es;
    zt += et;
    acc = fmaf(et, dt - ds, acc);
    s[k] = es;
    t[k] = et;
  }
  const float rzs = FAST ? fast_rcp_(zs) : 1.0f / zs;
  const float rzt = FAST ? fast_rcp_(zt) : 1.0f / zt;
  const float kl = (acc * rzt * invT + (logf(zs) - logf(zt))) * ((T * T) / (float)K);
#pragma unroll
  for (int k = 0; k < K; ++k) s[k] = fmaf(s[k], rzs, -(t[k] * rzt));
  return kl;
}

// Softmax over one 17-bin side + its expectation (Integral, gfl_head.py:32-44).
template <int K>
LD_HD float softmax_expect(const float* s, float* p /* [K] out */) {
  float m = s[0];
#pragma unroll
  for (int k = 1; k < K; ++k) m = fmaxf_(m, s[k]);
  float z = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    p[k] = expf(s[k] - m);
    z += p[k];
  }
  float rz = 1.0f / z, e = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    p[k] *= rz;
    e += p[k] * (float)k;
  }
  return e;
}

// Distribution focal loss of one side given its softmax p and log-sum-exp
// pieces (gfocal_loss.py:53-74): y in [0, reg_max - 0.1].
//   loss = CE(s, yl) * (yl + 1 - y) + CE(s, yl + 1) * (y - yl)
//   dloss/ds_k = p_k - wl [k == yl] - wr [k == yl + 1]      (wl + wr = 1)
template <int K>
LD_HD float dfl_side(const float* s, const float* p, float y, float* wl_out,
                     float* wr_out, int* yl_out) {
  int yl = (int)y;  // label.long(): truncation, y >= 0
  float wl = (float)(yl + 1) - y;
  float wr = y - (float)yl;
  // log softmax at yl and yl+1:  s_k - m - log z  ==  log p_k
  float m = s[0];
#pragma unroll
  for (int k = 1; k < K; ++k) m = fmaxf_(m, s[k]);
  float z = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) z += expf(s[k] - m);
  float lz = logf(z);
  float ll = 0.0f, lr = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    float lp = s[k] - m - lz;
    ll = (k == yl) ? lp : ll;
    lr = (k == yl + 1) ? lp : lr;
  }
  *wl_out = wl;
  *wr_out = wr;
  *yl_out = yl;
  (void)p;
  return -ll * wl - lr * wr;
}

// softmax_expect + dfl_side fused and IN PLACE (the same op sequence, so the same
// bits): on return s[k] = softmax(s)_k, *e_out = the Integral expectation, and
// the DFL value of target y is returned.  34 -> 17 live values for the dense
// sweep's rare positive anchors.
template <int K>
LD_HD float softmax_dfl_inplace(float* s, float y, float* e_out, float* wl_out,
                                float* wr_out, int* yl_out) {
  const int yl = (int)y;
  const float wl = (float)(yl + 1) - y, wr = y - (float)yl;
  float m = s[0];
#pragma unroll
  for (int k = 1; k < K; ++k) m = fmaxf_(m, s[k]);
  float z = 0.0f, sl = 0.0f, sr = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    sl = (k == yl) ? s[k] : sl;
    sr = (k == yl + 1) ? s[k] : sr;
    s[k] = expf(s[k] - m);
    z += s[k];
  }
  const float lz = logf(z), rz = 1.0f / z;
  float e = 0.0f;
#pragma unroll
  for (int k = 0; k < K; ++k) {
    s[k] *= rz;
    e += s[k] * (float)k;
  }
  // yl + 1 <= K - 1 (y <= reg_max - 0.1); a bin outside [0, K) contributes
  // log p = 0 exactly as dfl_side's unmatched select does
  const float ll = (yl >= 0 && yl < K) ? sl - m - lz : 0.0f;
  const float lr = (yl + 1 >= 0 && yl + 1 < K) ? sr - m - lz : 0.0f;
  *e_out = e;
  *wl_out = wl;
  *wr_out = wr;
  *yl_out = yl;
  return -ll * wl - lr * wr;
}

// bbox2distance clamp (core/bbox/transforms.py:171-180), max_dis = reg_max,
// eps = 0.1
LD_HD float clamp_dist(float d, float reg_max) {
  return fminf_(fmaxf_(d, 0.0f), reg_max - 0.1f);
}

}  // namespace ld
