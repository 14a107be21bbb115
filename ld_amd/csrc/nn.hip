// The non-GEMM layers of the LD train step on gfx950 -- all HBM-bound, so the
// rules are: one pass per tensor, float4 where the row length allows,
// per-channel / per-group reductions written as fixed-order two-stage sums
// (deterministic), statistics accumulated in fp64.
//
// Replaces (reference, mmdet 2.10 fork / mmcv / torch):
//   BatchNorm2d in eval mode with trainable affine   resnet.py:639-648
//       (norm_eval=True: running stats frozen, gamma/beta still learn)
//   ReLU / residual add                               resnet.py:65-92,260-299
//   GroupNorm(32) + ReLU of the head towers           gfl_head.py:102-126
//   MaxPool2d(3, 2, 1)                                resnet.py:570
//   FPN top-down nearest upsample + add               fpn.py:182-191
//   mmcv Scale                                        gfl_head.py:131-133,182
//   SGD(momentum, weight_decay)                       apis/train.py:88
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "ld_launch.h"

#include "../../include/ld_hip.h"

namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// block-wide sum of two doubles (256 threads), result valid in thread 0
__device__ __forceinline__ void block_sum2(double& a, double& b) {
  __shared__ double sa[4], sb[4];
  a = wave_sum_d(a);
  b = wave_sum_d(b);
  const int w = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sa[w] = a;
    sb[w] = b;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    a = sa[0] + sa[1] + sa[2] + sa[3];
    b = sb[0] + sb[1] + sb[2] + sb[3];
  }
  __syncthreads();
}

// ------------------------------------------------------------ BN (eval) ----
// scale = gamma * rsqrt(var + eps); shift = beta - mean * scale
__global__ void bn_prepare_kernel(const float* __restrict__ gamma,
                                  const float* __restrict__ beta,
                                  const float* __restrict__ mean,
                                  const float* __restrict__ var, float eps, int C,
                                  float* __restrict__ scale,
                                  float* __restrict__ shift,
                                  float* __restrict__ rstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float r = 1.0f / sqrtf(var[c] + eps);
  const float s = gamma[c] * r;
  scale[c] = s;
  shift[c] = beta[c] - mean[c] * s;
  if (rstd) rstd[c] = r;
}

// y = act(x * scale[c] + shift[c] (+ residual)); rows = N*C, row length P
template <bool VEC>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ residual,
    const float* __restrict__ scale, const float* __restrict__ shift, int C, int P,
    int relu, float* __restrict__ y) {
  const int row = blockIdx.y;
  const int c = row % C;
  const float s = scale[c], b = shift[c];
  const size_t base = (size_t)row * P;
  if (VEC) {
    const int i = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= P) return;
    float4 v = *reinterpret_cast<const float4*>(x + base + i);
    v.x = v.x * s + b; v.y = v.y * s + b; v.z = v.z * s + b; v.w = v.w * s + b;
    if (residual) {
      const float4 r = *reinterpret_cast<const float4*>(residual + base + i);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<float4*>(y + base + i) = v;
  } else {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= P) return;
    float v = x[base + i] * s + b;
    if (residual) v += residual[base + i];
    if (relu) v = fmaxf(v, 0.f);
    y[base + i] = v;
  }
}

// backward of the above.  dz = relu ? dy*(y>0) : dy;  dx = dz*scale[c];
// dres = dz (optional);  partial[c][split] = (sum dz, sum dz * xhat)
template <bool VEC>
__global__ __launch_bounds__(256) void bn_act_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ mean, const float* __restrict__ rstd, int N, int C,
    int P, int relu, float* __restrict__ dx, float* __restrict__ dres,
    double* __restrict__ partial) {
  const int c = blockIdx.x, split = blockIdx.y, nsplit = gridDim.y;
  const float s = scale[c], mu = mean ? mean[c] : 0.f, rs = rstd ? rstd[c] : 0.f;
  const long long total = (long long)N * P;
  long long per = (total + nsplit - 1) / nsplit;
  if (VEC) per = (per + 3) / 4 * 4;  // P % 4 == 0: slices start on float4 bounds
  const long long beg = (long long)split * per, end = min(total, beg + per);
  double s1 = 0.0, s2 = 0.0;
  if (VEC) {
    // walk the images the slice touches; inside an image the row is contiguous
    for (int n = (int)(beg / P); beg < end && n <= (int)((end - 1) / P); ++n) {
      const int p0 = (int)max(beg - (long long)n * P, 0LL);
      const int p1 = (int)min(end - (long long)n * P, (long long)P);
      const size_t base = ((size_t)n * C + c) * P;
      // Round 6: the loads of FOUR iterations are issued before the first use (up
      // to 12 x 16 bytes in flight per thread).  The loop used to wait for each
      // iteration's three loads in turn: 2.1 TB/s, 47 us per launch, 1.55 ms per C2
      // step on the critical stream.  Same per-thread accumulation order.
      const float* yq = relu ? y : dy;
      const float* xq = partial ? x : dy;
      for (int pb = p0 + 4 * threadIdx.x; pb < p1; pb += 4 * 1024) {
        float4 g4[4], y4[4], x4[4];
        bool ok[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int p = pb + u * 1024;
          ok[u] = p < p1;
          const size_t idx = base + (ok[u] ? p : pb);
          // unconditional loads (an absent operand aliases dy): a branch around a
          // load costs a vmcnt(0) at its join and the loop is serial again
          g4[u] = *reinterpret_cast<const float4*>(dy + idx);
          y4[u] = *reinterpret_cast<const float4*>(yq + idx);
          x4[u] = *reinterpret_cast<const float4*>(xq + idx);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          if (!ok[u]) continue;
          const size_t idx = base + pb + u * 1024;
          float dz[4] = {g4[u].x, g4[u].y, g4[u].z, g4[u].w};
          if (relu) {
            if (!(y4[u].x > 0.f)) dz[0] = 0.f;
            if (!(y4[u].y > 0.f)) dz[1] = 0.f;
            if (!(y4[u].z > 0.f)) dz[2] = 0.f;
            if (!(y4[u].w > 0.f)) dz[3] = 0.f;
          }
          if (dx)
            *reinterpret_cast<float4*>(dx + idx) =
                make_float4(dz[0] * s, dz[1] * s, dz[2] * s, dz[3] * s);
          if (dres)
            *reinterpret_cast<float4*>(dres + idx) =
                make_float4(dz[0], dz[1], dz[2], dz[3]);
          if (partial) {
            const float xv[4] = {x4[u].x, x4[u].y, x4[u].z, x4[u].w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              s1 += (double)dz[k];
              s2 += (double)(dz[k] * ((xv[k] - mu) * rs));
            }
          }
        }
      }
    }
  } else {
    for (long long e = beg + threadIdx.x; e < end; e += 256) {
      const int n = (int)(e / P);
      const int p = (int)(e - (long long)n * P);
      const size_t idx = ((size_t)n * C + c) * P + p;
      float dz = dy[idx];
      if (relu && !(y[idx] > 0.f)) dz = 0.f;
      if (dx) dx[idx] = dz * s;
      if (dres) dres[idx] = dz;
      if (partial) {
        s1 += (double)dz;
        s2 += (double)(dz * ((x[idx] - mu) * rs));
      }
    }
  }
  if (partial) {
    block_sum2(s1, s2);
    if (threadIdx.x == 0) {
      partial[((size_t)c * nsplit + split) * 2 + 0] = s1;
      partial[((size_t)c * nsplit + split) * 2 + 1] = s2;
    }
  }
}

// 16 lanes per channel: lane j adds slots j, j + 16, ... in order, then a fixed
// 4-step butterfly -- the same summation order on every run.  (One thread per
// channel walked up to 256 dependent 16-byte loads: 13 us per launch, 42
// launches per step.)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    const double* __restrict__ partial, int C, int nsplit, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int accumulate) {
  const int j = threadIdx.x & 15;
  const int c = blockIdx.x * 16 + (threadIdx.x >> 4);
  double s1 = 0.0, s2 = 0.0;
  if (c < C) {
    const double2* row = reinterpret_cast<const double2*>(partial) + (size_t)c * nsplit;
    for (int k = j; k < nsplit; k += 16) {
      const double2 v = row[k];
      s1 += v.x;
      s2 += v.y;
    }
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    s1 += __shfl_xor(s1, m, 16);
    s2 += __shfl_xor(s2, m, 16);
  }
  if (c >= C || j != 0) return;
  if (dbeta) dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
  if (dgamma) dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
}

// per-channel sum of dy over (N, P): conv bias gradient
__global__ __launch_bounds__(256) void bias_grad_kernel(
    const float* __restrict__ dy, int N, int C, int P, float* __restrict__ db,
    int accumulate) {
  const int c = blockIdx.x;
  double s = 0.0, z = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* r = dy + ((size_t)n * C + c) * P;
    for (int p = threadIdx.x; p < P; p += 256) s += (double)r[p];
  }
  block_sum2(s, z);
  if (threadIdx.x == 0) db[c] = accumulate ? db[c] + (float)s : (float)s;
}

// The same sum as per-channel partials for the deferred finalisation (round 5):
// grid (C, nsplit), split s owns the flat (n, p) range [s, s + 1) * chunk of the
// channel, 16-byte loads where the row allows; slot = (sum, 0) in the [C][nsplit]
// pair layout ld_bn_bwd_finalize_batch sums (a job with dgamma == NULL).  The
// one-block-per-channel kernel above moved 34 MB of the 100 x 168 level at 0.6 TB/s.
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(
    const float* __restrict__ dy, int N, int C, int P, int nsplit,
    double* __restrict__ partial) {
  const int c = blockIdx.x, sp = blockIdx.y;
  const long long total = (long long)N * P;
  const long long chunk = ((total + nsplit - 1) / nsplit + 3) & ~3LL;
  const long long beg = sp * chunk;
  long long end = beg + chunk;
  if (end > total) end = total;
  double s = 0.0, z = 0.0;
  const bool vec = P % 4 == 0 && ((uintptr_t)dy & 15) == 0;
  if (vec) {
    // P % 4 == 0 and chunk % 4 == 0: a float4 never straddles an image row
    for (long long q = beg + 4 * (long long)threadIdx.x; q < end; q += 1024) {
      const long long n = q / P, p = q - n * P;
      const float4 v = *reinterpret_cast<const float4*>(dy + ((size_t)n * C + c) * P + p);
      s += ((double)v.x + (double)v.y) + ((double)v.z + (double)v.w);
    }
  } else {
    for (long long q = beg + threadIdx.x; q < end; q += 256) {
      const long long n = q / P, p = q - n * P;
      s += (double)dy[((size_t)n * C + c) * P + p];
    }
  }
  block_sum2(s, z);
  if (threadIdx.x == 0) {
    partial[((size_t)c * nsplit + sp) * 2 + 0] = s;
    partial[((size_t)c * nsplit + sp) * 2 + 1] = 0.0;
  }
}

// -------------------------------------------------------------- GroupNorm ---
struct Levels {
  int num_levels;
  int P;  // positions per (n, c) row
  int off[LD_MAX_LEVELS + 1];
};

__device__ __forceinline__ int level_of_pos(const Levels& lv, int p) {
  int l = 0;
#pragma unroll
  for (int i = 1; i < LD_MAX_LEVELS; ++i)
    if (i < lv.num_levels && p >= lv.off[i]) l = i;
  return l;
}

// mean / rstd over (C/G channels) x A_l per (n, group, level), two stages so
// the 16800-cell level does not serialise on one workgroup:
//   stage 1  kGnSplit blocks per (n, g, l) -> partial (sum, sumsq) in fp64
//   stage 2  one thread per (n, g, l) adds the partials in fixed order
constexpr int kGnSplit = 32;
// positions per slice of a level with A positions, and how many slices hold any
__host__ __device__ __forceinline__ int gn_slice_len(int A) {
  const int per = (A + kGnSplit - 1) / kGnSplit;
  return per > 1024 ? per : 1024;
}
__host__ __device__ __forceinline__ int gn_slices(int A) {
  const int per = gn_slice_len(A);
  return A > 0 ? (A + per - 1) / per : 0;
}

__global__ __launch_bounds__(256) void gn_stats_partial_kernel(
    const float* __restrict__ x, Levels lv, int C, int G,
    double* __restrict__ partial) {
  // Only the slices that own positions are launched (gn_slices(): 26 of the
  // 5 x 32 per (n, group) at the C2 pyramid).  Round 3 first launched all of
  // them and let the empty ones return at once: 10 240 workgroups, 8 576 of them
  // empty, cost more to dispatch than the 1 664 real ones to run (27 us per
  // launch for 46 MB).
  const int L = lv.num_levels;
  int per_ng = 0;
#pragma unroll
  for (int i = 0; i < LD_MAX_LEVELS; ++i)
    if (i < L) per_ng += gn_slices(lv.off[i + 1] - lv.off[i]);
  const int ng = blockIdx.x / per_ng;
  int sp = blockIdx.x - ng * per_ng, l = 0;
#pragma unroll
  for (int i = 0; i < LD_MAX_LEVELS - 1; ++i) {
    const int c = i < L ? gn_slices(lv.off[i + 1] - lv.off[i]) : 0;
    if (l == i && i + 1 < L && sp >= c) {
      sp -= c;
      l = i + 1;
    }
  }
  const int g = ng % G, n = ng / G;
  const int ngl = (n * G + g) * L + l;
  const int cpg = C / G, A = lv.off[l + 1] - lv.off[l];
  const float* base = x + ((size_t)n * C + (size_t)g * cpg) * lv.P + lv.off[l];
  // split sp owns positions [beg, end) of the level in every channel of the
  // group: coalesced rows, no per-element index arithmetic; at least 1024
  // positions per slice (the small levels are one workgroup each)
  const int per = gn_slice_len(A);
  const int beg = sp * per, end = min(A, beg + per);
  double* const slot = partial + ((size_t)ngl * kGnSplit + sp) * 2;
  double s = 0.0, q = 0.0;
  // 16-byte loads on the aligned body of the slice (round 3: the scalar loop ran
  // at 1.8 TB/s), scalar on its <= 3-cell edges
  const bool al = ((uintptr_t)base % 16 == 0) && (lv.P % 4 == 0);
  const int b0 = al ? min((beg + 3) & ~3, end) : end;
  const int b1 = al ? max(end & ~3, b0) : end;
  for (int p = b0 + 4 * (int)threadIdx.x; p < b1; p += 1024) {
#pragma unroll 8
    for (int ch = 0; ch < cpg; ++ch) {
      const float4 t = *reinterpret_cast<const float4*>(base + (size_t)ch * lv.P + p);
      const double v0 = t.x, v1 = t.y, v2 = t.z, v3 = t.w;
      s += (v0 + v1) + (v2 + v3);
      q += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
    }
  }
  for (int p = beg + threadIdx.x; p < end; p += 256) {
    if (p >= b0 && p < b1) continue;
#pragma unroll 8
    for (int ch = 0; ch < cpg; ++ch) {
      const double v = (double)base[(size_t)ch * lv.P + p];
      s += v;
      q += v * v;
    }
  }
  block_sum2(s, q);
  if (threadIdx.x == 0) {
    slot[0] = s;
    slot[1] = q;
  }
}

__global__ void gn_stats_final_kernel(const double* __restrict__ partial, Levels lv,
                                      int N, int C, int G, float eps,
                                      float* __restrict__ mean,
                                      float* __restrict__ rstd) {
  const int L = lv.num_levels;
  const int ngl = blockIdx.x * blockDim.x + threadIdx.x;
  if (ngl >= N * G * L) return;
  const int l = ngl % L;
  const double total = (double)(C / G) * (lv.off[l + 1] - lv.off[l]);
  double s = 0.0, q = 0.0;
  const int nk = gn_slices(lv.off[l + 1] - lv.off[l]);  // the slices that were written
  for (int k = 0; k < nk; ++k) {
    s += partial[((size_t)ngl * kGnSplit + k) * 2 + 0];
    q += partial[((size_t)ngl * kGnSplit + k) * 2 + 1];
  }
  const double m = s / total;
  double var = q / total - m * m;
  if (var < 0.0) var = 0.0;
  mean[ngl] = (float)m;
  rstd[ngl] = (float)(1.0 / sqrt(var + (double)eps));
}

// y = relu?( (x - mean) * rstd * gamma[c] + beta[c] )
// y = xhat * gamma + beta, the product-sum PINNED as one fma.  Round 6: the lean
// backward (ld_gn_backward_c8_lean) recomputes the ReLU mask y > 0 from x with this
// same function instead of reading y back -- forward and backward must round the
// same way whatever the compiler's contraction choice is at either site.
__device__ __forceinline__ float gn_affine(float in, float mu, float rs, float ga,
                                           float be) {
  return __builtin_fmaf((in - mu) * rs, ga, be);
}

__global__ __launch_bounds__(256) void gn_apply_kernel(
    const float* __restrict__ x, Levels lv, int C, int G,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
    float* __restrict__ y) {
  const int row = blockIdx.y;  // n*C + c
  const int c = row % C, n = row / C;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= lv.P) return;
  const int l = level_of_pos(lv, p);
  const size_t o = ((size_t)n * G + c / (C / G)) * lv.num_levels + l;
  const size_t idx = (size_t)row * lv.P + p;
  float v = gn_affine(x[idx], mean[o], rstd[o], gamma[c], beta[c]);
  if (relu) v = fmaxf(v, 0.f);
  y[idx] = v;
}

// the same, four consecutive positions per thread (rows 16-byte aligned)
__global__ __launch_bounds__(256) void gn_apply4_kernel(
    const float* __restrict__ x, Levels lv, int C, int G,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
    float* __restrict__ y) {
  const int row = blockIdx.y;  // n*C + c
  const int c = row % C, n = row / C;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= lv.P) return;
  const size_t ob = ((size_t)n * G + c / (C / G)) * lv.num_levels;
  const size_t idx = (size_t)row * lv.P + p;
  const float4 v = *reinterpret_cast<const float4*>(x + idx);
  const float ga = gamma[c], be = beta[c];
  const int l0 = level_of_pos(lv, p), l3 = level_of_pos(lv, p + 3);
  float in[4] = {v.x, v.y, v.z, v.w}, out[4];
  if (l0 == l3) {
    const float mu = mean[ob + l0], rs = rstd[ob + l0];
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = gn_affine(in[k], mu, rs, ga, be);
  } else {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int l = level_of_pos(lv, p + k);
      out[k] = gn_affine(in[k], mean[ob + l], rstd[ob + l], ga, be);
    }
  }
  if (relu) {
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = fmaxf(out[k], 0.f);
  }
  *reinterpret_cast<float4*>(y + idx) = make_float4(out[0], out[1], out[2], out[3]);
}

// ---- bf16 channel-blocked side output ("C8", conv_bf16.hip) ------------------
// In bf16 mode the conv that consumes y (forward) or dx (data gradient) wants
// its activation operand as (N, C/8, P, 8) bf16.  These variants own 8 channels
// x 4 positions per thread, write the SAME fp32 tensor as the kernels above
// (identical expressions) and, from the values already in registers, the C8
// image: no separate conversion launch, no re-read.
typedef float gn_floatx8 __attribute__((ext_vector_type(8)));
typedef __bf16 gn_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gn_uintx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void store_c8x4(gn_uintx4* __restrict__ dst,
                                           const float (&v)[8][4]) {
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    gn_floatx8 f;
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = v[e][k];
    dst[k] = __builtin_bit_cast(gn_uintx4, __builtin_convertvector(f, gn_bf16x8));
  }
}

// y = act(x * scale[c] + shift[c] (+ residual)) with the C8 side output
__global__ __launch_bounds__(256) void bn_act_fwd_c8_kernel(
    const float* __restrict__ x, const float* __restrict__ residual,
    const float* __restrict__ scale, const float* __restrict__ shift, int C, int P,
    int relu, float* __restrict__ y, gn_uintx4* __restrict__ y_c8) {
  const int C8 = C >> 3;
  const int blk = blockIdx.y;  // n * C8 + c8
  const int c8 = blk % C8, n = blk / C8;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= P) return;
  float out[8][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    const float s = scale[c], b = shift[c];
    const size_t idx = ((size_t)n * C + c) * P + p;
    float4 v = *reinterpret_cast<const float4*>(x + idx);
    v.x = v.x * s + b; v.y = v.y * s + b; v.z = v.z * s + b; v.w = v.w * s + b;
    if (residual) {
      const float4 r = *reinterpret_cast<const float4*>(residual + idx);
      v.x += r.x; v.y += r.y; v.z += r.z; v.w += r.w;
    }
    if (relu) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f);
      v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<float4*>(y + idx) = v;
    out[e][0] = v.x; out[e][1] = v.y; out[e][2] = v.z; out[e][3] = v.w;
  }
  store_c8x4(y_c8 + (size_t)blk * P + p, out);
}

// backward of bn_act with the C8 image of dx as a side output (round 3): in
// bf16 mode dx goes straight into the conv's data- and weight-gradient kernels,
// which take C8 operands -- round 2 converted it with a separate launch per conv
// (~42 to_c8 launches per step).  Thread = 8 channels x 4 positions; the same
// expressions as bn_act_bwd_kernel; per-channel (sum dz, sum dz*xhat) partials
// per workgroup, one slot per (image, position block), summed in fixed order by
// bn_bwd_finalize_kernel.
__global__ __launch_bounds__(64) void bn_act_bwd_c8_kernel(
    const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ x, const float* __restrict__ scale,
    const float* __restrict__ mean, const float* __restrict__ rstd, int C, int P,
    int relu, float* __restrict__ dx, float* __restrict__ dres,
    gn_uintx4* __restrict__ dx_c8, double* __restrict__ partial, int nslots) {
  const int C8 = C >> 3;
  const int blk = blockIdx.y;  // n * C8 + c8
  const int c8 = blk % C8, n = blk / C8;
  // one wavefront per workgroup: 256 positions x 8 channels.  (Round 3 first
  // ran 256-thread workgroups: 320 of them for a 256-channel 50x84 map -- 34 us
  // against 24 us of the per-channel kernel it replaces.)
  const int p = (blockIdx.x * 64 + threadIdx.x) * 4;
  const bool live = p < P;
  float out[8][4];
  double s1[8], s2[8];
  // Round 6: every load of the thread -- dy, y, x of its eight channels, 24 x 16
  // bytes -- is issued before the first use.  The per-channel loop used to wait for
  // its own three loads each time round (eight dependent memory latencies per wave:
  // 2.9 TB/s, 1.26 ms per C2 step on the critical stream).  A lane past the end of
  // the row reads the row's first cells and discards them.  Same expressions.
  float4 g8[8], y8[8], x8[8];
  const int pl = live ? p : 0;
  const float* yq = relu ? y : dy;
  const float* xq = partial ? x : dy;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const size_t idx = ((size_t)n * C + c8 * 8 + e) * P + pl;
    // unconditional (an absent operand aliases dy): no branch, no vmcnt(0) joins
    g8[e] = *reinterpret_cast<const float4*>(dy + idx);
    y8[e] = *reinterpret_cast<const float4*>(yq + idx);
    x8[e] = *reinterpret_cast<const float4*>(xq + idx);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    const float s = scale[c];
    s1[e] = 0.0;
    s2[e] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; ++k) out[e][k] = 0.0f;
    if (!live) continue;
    const size_t idx = ((size_t)n * C + c) * P + p;
    float dz[4] = {g8[e].x, g8[e].y, g8[e].z, g8[e].w};
    if (relu) {
      if (!(y8[e].x > 0.f)) dz[0] = 0.f;
      if (!(y8[e].y > 0.f)) dz[1] = 0.f;
      if (!(y8[e].z > 0.f)) dz[2] = 0.f;
      if (!(y8[e].w > 0.f)) dz[3] = 0.f;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) out[e][k] = dz[k] * s;
    if (dx)
      *reinterpret_cast<float4*>(dx + idx) =
          make_float4(out[e][0], out[e][1], out[e][2], out[e][3]);
    if (dres)
      *reinterpret_cast<float4*>(dres + idx) = make_float4(dz[0], dz[1], dz[2], dz[3]);
    if (partial) {
      const float mu = mean[c], rs = rstd[c];
      const float xv[4] = {x8[e].x, x8[e].y, x8[e].z, x8[e].w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s1[e] += (double)dz[k];
        s2[e] += (double)(dz[k] * ((xv[k] - mu) * rs));
      }
    }
  }
  if (live) store_c8x4(dx_c8 + (size_t)blk * P + p, out);
  if (partial) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double a = wave_sum_d(s1[e]), b = wave_sum_d(s2[e]);
      if (threadIdx.x == 0) {
        const int slot = n * gridDim.x + blockIdx.x;
        const size_t at = ((size_t)(c8 * 8 + e) * nslots + slot) * 2;
        partial[at + 0] = a;
        partial[at + 1] = b;
      }
    }
  }
}

// Round 6, the lean BN backward of bf16 mode: the two saved activations arrive as
// the bf16 C8 images the forward wrote anyway -- y (only its sign is used: the ReLU
// mask, identical to the fp32 test) and the conv result x before the affine (for
// d(gamma); the conv epilogue writes it as a C8 image instead of fp32,
// ld_conv_epilogue_t.y_raw_c8).  8 instead of 12 bytes read per element, and the
// forward wrote 2 instead of 4 for x.  Same thread shape and the same expressions
// as bn_act_bwd_c8_kernel; a thread's 8 channels x 4 positions of an image are 64
// contiguous bytes.
template <int V>  // positions per thread: 4 (P % 4 == 0) or 2 (P % 2 == 0: 25 x 42)
__global__ __launch_bounds__(64) void bn_act_bwd_c8in_kernel(
    const float* __restrict__ dy, const gn_uintx4* __restrict__ y_img,
    const gn_uintx4* __restrict__ x_img, const float* __restrict__ scale,
    const float* __restrict__ mean, const float* __restrict__ rstd, int C, int P,
    int relu, float* __restrict__ dx, float* __restrict__ dres,
    gn_uintx4* __restrict__ dx_c8, double* __restrict__ partial, int nslots) {
  typedef float fvec __attribute__((ext_vector_type(V)));
  const int C8 = C >> 3;
  const int blk = blockIdx.y;  // n * C8 + c8
  const int c8 = blk % C8, n = blk / C8;
  const int p = (blockIdx.x * 64 + threadIdx.x) * V;
  const bool live = p < P;
  const int pl = live ? p : 0;
  float out[8][V];
  double s1[8], s2[8];
  fvec g8[8];
  gn_uintx4 yq[V], xq[V];
  const gn_uintx4* yp = relu ? y_img : x_img;  // unconditional loads, no branch
#pragma unroll
  for (int e = 0; e < 8; ++e)
    g8[e] = *reinterpret_cast<const fvec*>(dy + ((size_t)n * C + c8 * 8 + e) * P + pl);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    yq[k] = yp[(size_t)blk * P + pl + k];
    xq[k] = x_img[(size_t)blk * P + pl + k];
  }
  gn_floatx8 yf[V], xf[V];
#pragma unroll
  for (int k = 0; k < V; ++k) {
    yf[k] = __builtin_convertvector(__builtin_bit_cast(gn_bf16x8, yq[k]), gn_floatx8);
    xf[k] = __builtin_convertvector(__builtin_bit_cast(gn_bf16x8, xq[k]), gn_floatx8);
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    const float s = scale[c];
    s1[e] = 0.0;
    s2[e] = 0.0;
#pragma unroll
    for (int k = 0; k < V; ++k) out[e][k] = 0.0f;
    if (!live) continue;
    const size_t idx = ((size_t)n * C + c) * P + p;
    fvec dz = g8[e];
    if (relu) {
#pragma unroll
      for (int k = 0; k < V; ++k)
        if (!(yf[k][e] > 0.f)) dz[k] = 0.f;
    }
    fvec o;
#pragma unroll
    for (int k = 0; k < V; ++k) {
      out[e][k] = dz[k] * s;
      o[k] = out[e][k];
    }
    if (dx) *reinterpret_cast<fvec*>(dx + idx) = o;
    if (dres) *reinterpret_cast<fvec*>(dres + idx) = dz;
    if (partial) {
      const float mu = mean[c], rs = rstd[c];
#pragma unroll
      for (int k = 0; k < V; ++k) {
        s1[e] += (double)dz[k];
        s2[e] += (double)(dz[k] * ((xf[k][e] - mu) * rs));
      }
    }
  }
  if (live) {
#pragma unroll
    for (int k = 0; k < V; ++k) {
      gn_floatx8 f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = out[e][k];
      dx_c8[(size_t)blk * P + p + k] =
          __builtin_bit_cast(gn_uintx4, __builtin_convertvector(f, gn_bf16x8));
    }
  }
  if (partial) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const double a = wave_sum_d(s1[e]), b = wave_sum_d(s2[e]);
      if (threadIdx.x == 0) {
        const int slot = n * gridDim.x + blockIdx.x;
        const size_t at = ((size_t)(c8 * 8 + e) * nslots + slot) * 2;
        partial[at + 0] = a;
        partial[at + 1] = b;
      }
    }
  }
}

__global__ __launch_bounds__(256) void gn_apply_c8_kernel(
    const float* __restrict__ x, Levels lv, int C, int G,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
    float* __restrict__ y, gn_uintx4* __restrict__ y_c8) {
  const int C8 = C >> 3;
  const int blk = blockIdx.y;  // n * C8 + c8
  const int c8 = blk % C8, n = blk / C8;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= lv.P) return;
  const int l0 = level_of_pos(lv, p), l3 = level_of_pos(lv, p + 3);
  float out[8][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    const size_t ob = ((size_t)n * G + c / (C / G)) * lv.num_levels;
    const size_t idx = ((size_t)n * C + c) * lv.P + p;
    const float4 v = *reinterpret_cast<const float4*>(x + idx);
    const float ga = gamma[c], be = beta[c];
    const float in[4] = {v.x, v.y, v.z, v.w};
    if (l0 == l3) {
      const float mu = mean[ob + l0], rs = rstd[ob + l0];
#pragma unroll
      for (int k = 0; k < 4; ++k) out[e][k] = gn_affine(in[k], mu, rs, ga, be);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int l = level_of_pos(lv, p + k);
        out[e][k] = gn_affine(in[k], mean[ob + l], rstd[ob + l], ga, be);
      }
    }
    if (relu) {
#pragma unroll
      for (int k = 0; k < 4; ++k) out[e][k] = fmaxf(out[e][k], 0.f);
    }
    if (y)  // null: the C8 image is the only output (ld_gn_forward_c8, y == NULL)
      *reinterpret_cast<float4*>(y + idx) =
          make_float4(out[e][0], out[e][1], out[e][2], out[e][3]);
  }
  store_c8x4(y_c8 + (size_t)blk * lv.P + p, out);
}

// Round 6: the per-(image, group, level) statistics a thread needs -- mean, rstd
// and the two group means of the gradient -- are the same for a whole workgroup
// row, so they are read ONCE per workgroup through scalar loads (uniform address)
// into a small table and selected by level with v_cndmask, with a fast path when a
// thread's four positions share a level (all but <= 4 threads of a row).  Rounds
// 2-5 fetched them per element with per-lane loads: 16 tiny gathers per 48 bytes
// of payload (15 / 120 vector loads per thread where 3 / 24 carry data).  Same
// expressions, same operand values: bit-identical outputs (LD_NN_OLD=1 keeps the
// old kernels for tests/test_gpu_layers.py::test_round6_norm_backward_kernels_bit_
// identical).  Measured in the step, same box, old vs new (with the BN backward
// change below): bf16 11.99 -> 11.85 ms, fp32 34.23 -> 34.20 ms -- in fp32 these
// HBM-bound launches run UNDER the other stream's MFMA-bound convolutions (their
// in-step durations, 199 us average against 65 us alone, are contention, not cost),
// which is why the fp32 step only moves with the conv kernels.
struct GnRowTab {
  float mu[LD_MAX_LEVELS], rs[LD_MAX_LEVELS], f1[LD_MAX_LEVELS], f2[LD_MAX_LEVELS];
};

__device__ __forceinline__ void gn_load_tab(GnRowTab& t, const float* __restrict__ mean,
                                            const float* __restrict__ rstd,
                                            const float* __restrict__ gm, size_t ob,
                                            int L) {
#pragma unroll
  for (int l = 0; l < LD_MAX_LEVELS; ++l) {
    const size_t o = ob + (l < L ? l : 0);  // wave-uniform address: scalar loads
    t.mu[l] = mean[o];
    t.rs[l] = rstd[o];
    t.f1[l] = gm[o * 2 + 0];
    t.f2[l] = gm[o * 2 + 1];
  }
}

__device__ __forceinline__ void gn_pick(const GnRowTab& t, int l, float& mu, float& rs,
                                        float& f1, float& f2) {
  mu = t.mu[0];
  rs = t.rs[0];
  f1 = t.f1[0];
  f2 = t.f2[0];
#pragma unroll
  for (int i = 1; i < LD_MAX_LEVELS; ++i)
    if (l == i) {
      mu = t.mu[i];
      rs = t.rs[i];
      f1 = t.f1[i];
      f2 = t.f2[i];
    }
}

// MASK_X (the lean backward, bf16 mode): the ReLU mask y > 0 is recomputed from x
// with the forward's own expression (gn_affine) instead of reading y -- 8 instead
// of 12 bytes read per element; dx may be null (only the C8 image is consumed).
template <bool MASK_X>
__global__ __launch_bounds__(256) void gn_bwd_apply_c8_kernel(
    const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ x, Levels lv, int C, int G,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ gm, int relu, float* __restrict__ dx,
    gn_uintx4* __restrict__ dx_c8) {
  const int C8 = C >> 3;
  const int blk = blockIdx.y;
  const int c8 = blk % C8, n = blk / C8;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= lv.P) return;
  const int L = lv.num_levels;
  const int cpg = C / G;
  const bool one_group = (cpg & 7) == 0;  // the 8 channels of a block share a group
  const int l0 = level_of_pos(lv, p), l3 = level_of_pos(lv, p + 3);
  GnRowTab tab;
  float mu0 = 0.f, rs0 = 0.f, f10 = 0.f, f20 = 0.f;
  if (one_group) {
    gn_load_tab(tab, mean, rstd, gm, ((size_t)n * G + (c8 * 8) / cpg) * L, L);
    gn_pick(tab, l0, mu0, rs0, f10, f20);
  }
  // every load of the thread first (24 x 16 bytes in flight)
  const float* yq = relu ? y : dy;
  float4 t0[8], t2[8], t1[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const size_t idx = ((size_t)n * C + c8 * 8 + e) * lv.P + p;
    t0[e] = *reinterpret_cast<const float4*>(dy + idx);
    t2[e] = *reinterpret_cast<const float4*>(x + idx);
    if (!MASK_X)
      t1[e] = *reinterpret_cast<const float4*>(yq + idx);  // unconditional: no branch
  }
  float out[8][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    const size_t idx = ((size_t)n * C + c) * lv.P + p;
    const float ga = gamma[c];
    float be = 0.f;
    if (MASK_X) be = beta[c];
    if (!one_group) {
      gn_load_tab(tab, mean, rstd, gm, ((size_t)n * G + c / cpg) * L, L);
      gn_pick(tab, l0, mu0, rs0, f10, f20);
    }
    const float a_dy[4] = {t0[e].x, t0[e].y, t0[e].z, t0[e].w};
    const float a_x[4] = {t2[e].x, t2[e].y, t2[e].z, t2[e].w};
    float a_y[4] = {1.f, 1.f, 1.f, 1.f};
    if (!MASK_X) {
      a_y[0] = t1[e].x; a_y[1] = t1[e].y; a_y[2] = t1[e].z; a_y[3] = t1[e].w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float mu = mu0, rs = rs0, fm1 = f10, fm2 = f20;
      if (l0 != l3) gn_pick(tab, level_of_pos(lv, p + k), mu, rs, fm1, fm2);
      float dz = a_dy[k];
      if (MASK_X) a_y[k] = gn_affine(a_x[k], mu, rs, ga, be);
      if (relu && !(a_y[k] > 0.f)) dz = 0.f;
      const float xh = (a_x[k] - mu) * rs;
      out[e][k] = rs * (ga * dz - fm1 - xh * fm2);
    }
    if (!MASK_X || dx)
      *reinterpret_cast<float4*>(dx + idx) =
          make_float4(out[e][0], out[e][1], out[e][2], out[e][3]);
  }
  store_c8x4(dx_c8 + (size_t)blk * lv.P + p, out);
}

namespace old_r5 {
__global__ __launch_bounds__(256) void gn_bwd_apply_c8_kernel(
    const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ x, Levels lv, int C, int G,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ gm, int relu,
    float* __restrict__ dx, gn_uintx4* __restrict__ dx_c8) {
  const int C8 = C >> 3;
  const int blk = blockIdx.y;
  const int c8 = blk % C8, n = blk / C8;
  const int p = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (p >= lv.P) return;
  const int L = lv.num_levels;
  float out[8][4];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = c8 * 8 + e;
    const size_t ob = ((size_t)n * G + c / (C / G)) * L;
    const size_t idx = ((size_t)n * C + c) * lv.P + p;
    const float ga = gamma[c];
    const float4 t0 = *reinterpret_cast<const float4*>(dy + idx);
    const float4 t2 = *reinterpret_cast<const float4*>(x + idx);
    const float a_dy[4] = {t0.x, t0.y, t0.z, t0.w};
    const float a_x[4] = {t2.x, t2.y, t2.z, t2.w};
    float a_y[4] = {1.f, 1.f, 1.f, 1.f};
    if (relu) {
      const float4 t1 = *reinterpret_cast<const float4*>(y + idx);
      a_y[0] = t1.x; a_y[1] = t1.y; a_y[2] = t1.z; a_y[3] = t1.w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int l = level_of_pos(lv, p + k);
      const float mu = mean[ob + l], rs = rstd[ob + l];
      const float fm1 = gm[(ob + l) * 2 + 0], fm2 = gm[(ob + l) * 2 + 1];
      float dz = a_dy[k];
      if (relu && !(a_y[k] > 0.f)) dz = 0.f;
      const float xh = (a_x[k] - mu) * rs;
      out[e][k] = rs * (ga * dz - fm1 - xh * fm2);
    }
    *reinterpret_cast<float4*>(dx + idx) =
        make_float4(out[e][0], out[e][1], out[e][2], out[e][3]);
  }
  store_c8x4(dx_c8 + (size_t)blk * lv.P + p, out);
}

}  // namespace old_r5

// backward pass A: per (n, c, level): s1 = sum dz, s2 = sum dz * xhat, each
// level cut into kGnBwdSplit slices (the 16800-cell level would otherwise sit
// on two workgroups per CU)
constexpr int kGnBwdSplit = 8;

__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(
    const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ x, Levels lv, int C, int G,
    const float* __restrict__ mean, const float* __restrict__ rstd, int relu,
    double* __restrict__ sums) {
  const int L = lv.num_levels;
  const int sp = blockIdx.x % kGnBwdSplit;
  const int rl = blockIdx.x / kGnBwdSplit;
  const int l = rl % L, row = rl / L;  // row = n*C + c
  const int c = row % C, n = row / C;
  const int A = lv.off[l + 1] - lv.off[l];
  const int per = max((A + kGnBwdSplit - 1) / kGnBwdSplit, 1024);
  const int beg = sp * per, end = min(A, beg + per);
  if (beg >= end) {
    if (threadIdx.x == 0) {
      sums[(size_t)blockIdx.x * 2 + 0] = 0.0;
      sums[(size_t)blockIdx.x * 2 + 1] = 0.0;
    }
    return;
  }
  const size_t o = ((size_t)n * G + c / (C / G)) * L + l;
  const float mu = mean[o], rs = rstd[o];
  const size_t base = (size_t)row * lv.P + lv.off[l];
  double s1 = 0.0, s2 = 0.0;
  for (int p = beg + threadIdx.x; p < end; p += 256) {
    float dz = dy[base + p];
    if (relu && !(y[base + p] > 0.f)) dz = 0.f;
    s1 += (double)dz;
    s2 += (double)(dz * ((x[base + p] - mu) * rs));
  }
  block_sum2(s1, s2);
  if (threadIdx.x == 0) {
    sums[(size_t)blockIdx.x * 2 + 0] = s1;
    sums[(size_t)blockIdx.x * 2 + 1] = s2;
  }
}

// Round 3: the same sums with ONE workgroup per (n, c) row walking all levels
// (16-byte loads on the aligned body of each level, the few edge cells scalar):
// 512 workgroups stream 270 KB each instead of 20 480 slices of <= 2100 cells
// whose cost was their own latency (62 us per launch, 1.5 TB/s).  Writes slice 0
// of every (row, level) directly -- no fold launch.  Per-thread accumulation
// order differs from the sliced kernel (fp64 sums: ~1e-16 relative).
template <bool MASK_X>
__global__ __launch_bounds__(256) void gn_bwd_reduce_row_kernel(
    const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ x, Levels lv, int C, int G,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ beta, int relu,
    double* __restrict__ sums) {
  const int L = lv.num_levels;
  const int row = blockIdx.x;  // n*C + c
  const int c = row % C, n = row / C;
  const size_t rbase = (size_t)row * lv.P;
  float ga = 0.f, be = 0.f;
  if (MASK_X) {  // the ReLU mask from x (gn_affine), not from y
    ga = gamma[c];
    be = beta[c];
  }
  for (int l = 0; l < L; ++l) {
    const size_t o = ((size_t)n * G + c / (C / G)) * L + l;
    const float mu = mean[o], rs = rstd[o];
    const int p0 = lv.off[l], p1 = lv.off[l + 1];
    // aligned body [b0, b1): rbase % 4 == 0 is guaranteed by the launcher
    const int b0 = min((p0 + 3) & ~3, p1), b1 = max(p1 & ~3, b0);
    double s1 = 0.0, s2 = 0.0;
    for (int p = b0 + 4 * (int)threadIdx.x; p < b1; p += 1024) {
      const size_t idx = rbase + p;
      const float4 g = *reinterpret_cast<const float4*>(dy + idx);
      const float4 xx = *reinterpret_cast<const float4*>(x + idx);
      float dz[4] = {g.x, g.y, g.z, g.w};
      const float xv[4] = {xx.x, xx.y, xx.z, xx.w};
      if (relu && MASK_X) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (!(gn_affine(xv[k], mu, rs, ga, be) > 0.f)) dz[k] = 0.f;
      } else if (relu) {
        const float4 yy = *reinterpret_cast<const float4*>(y + idx);
        if (!(yy.x > 0.f)) dz[0] = 0.f;
        if (!(yy.y > 0.f)) dz[1] = 0.f;
        if (!(yy.z > 0.f)) dz[2] = 0.f;
        if (!(yy.w > 0.f)) dz[3] = 0.f;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        s1 += (double)dz[k];
        s2 += (double)(dz[k] * ((xv[k] - mu) * rs));
      }
    }
    // edges: [p0, b0) and [b1, p1), at most 3 cells each
    const int ne = (b0 - p0) + (p1 - b1);
    if ((int)threadIdx.x < ne) {
      const int t = threadIdx.x;
      const int p = t < b0 - p0 ? p0 + t : b1 + (t - (b0 - p0));
      const size_t idx = rbase + p;
      float dz = dy[idx];
      const float xe = x[idx];
      if (relu && !((MASK_X ? gn_affine(xe, mu, rs, ga, be) : y[idx]) > 0.f)) dz = 0.f;
      s1 += (double)dz;
      s2 += (double)(dz * ((xe - mu) * rs));
    }
    block_sum2(s1, s2);
    if (threadIdx.x == 0) {
      const size_t at = ((size_t)row * L + l) * kGnBwdSplit * 2;
      sums[at + 0] = s1;
      sums[at + 1] = s2;
    }
  }
}

// slices -> per (n, c, level) sums (in place at slice 0), fixed order
__global__ void gn_bwd_fold_kernel(double* __restrict__ sums, int rows_levels) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows_levels) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < kGnBwdSplit; ++k) {
    s1 += sums[((size_t)i * kGnBwdSplit + k) * 2 + 0];
    s2 += sums[((size_t)i * kGnBwdSplit + k) * 2 + 1];
  }
  sums[(size_t)i * kGnBwdSplit * 2 + 0] = s1;
  sums[(size_t)i * kGnBwdSplit * 2 + 1] = s2;
}

// per (n, group, level): m1 = mean_group(gamma*dz), m2 = mean_group(gamma*dz*xhat)
__global__ void gn_bwd_group_kernel(const double* __restrict__ sums, Levels lv, int N,
                                    int C, int G, const float* __restrict__ gamma,
                                    float* __restrict__ gm) {
  const int L = lv.num_levels;
  const int ngl = blockIdx.x * blockDim.x + threadIdx.x;
  if (ngl >= N * G * L) return;
  const int l = ngl % L, g = (ngl / L) % G, n = ngl / (L * G);
  const int cpg = C / G, A = lv.off[l + 1] - lv.off[l];
  double m1 = 0.0, m2 = 0.0;
  for (int k = 0; k < cpg; ++k) {
    const int cc = g * cpg + k;
    const size_t r = ((size_t)(n * C + cc) * L + l) * kGnBwdSplit * 2;
    m1 += (double)gamma[cc] * sums[r + 0];
    m2 += (double)gamma[cc] * sums[r + 1];
  }
  const double cnt = (double)cpg * A;
  gm[(size_t)ngl * 2 + 0] = (float)(m1 / cnt);
  gm[(size_t)ngl * 2 + 1] = (float)(m2 / cnt);
}

// backward pass B: dx = rstd * (gamma*dz - m1 - xhat*m2)
template <int V>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ x, Levels lv, int C, int G,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ gm, int relu,
    float* __restrict__ dx) {
  const int row = blockIdx.y;
  const int c = row % C, n = row / C;
  const int p = (blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= lv.P) return;
  const int L = lv.num_levels, g = c / (C / G);
  const size_t idx = (size_t)row * lv.P + p;
  const float ga = gamma[c];
  GnRowTab tab;  // the row's statistics: scalar loads, once per workgroup (see above)
  gn_load_tab(tab, mean, rstd, gm, ((size_t)n * G + g) * L, L);
  float a_dy[V], a_y[V], a_x[V], out[V];
  if (V == 4) {
    const float4 t0 = *reinterpret_cast<const float4*>(dy + idx);
    const float4 t2 = *reinterpret_cast<const float4*>(x + idx);
    a_dy[0] = t0.x; a_dy[1] = t0.y; a_dy[2] = t0.z; a_dy[3] = t0.w;
    a_x[0] = t2.x; a_x[1] = t2.y; a_x[2] = t2.z; a_x[3] = t2.w;
    if (relu) {
      const float4 t1 = *reinterpret_cast<const float4*>(y + idx);
      a_y[0] = t1.x; a_y[1] = t1.y; a_y[2] = t1.z; a_y[3] = t1.w;
    }
  } else {
    a_dy[0] = dy[idx];
    a_x[0] = x[idx];
    if (relu) a_y[0] = y[idx];
  }
  const int l0 = level_of_pos(lv, p), l3 = level_of_pos(lv, p + V - 1);
  float mu0, rs0, f10, f20;
  gn_pick(tab, l0, mu0, rs0, f10, f20);
#pragma unroll
  for (int k = 0; k < V; ++k) {
    float mu = mu0, rs = rs0, fm1 = f10, fm2 = f20;
    if (l0 != l3) gn_pick(tab, level_of_pos(lv, p + k), mu, rs, fm1, fm2);
    float dz = a_dy[k];
    if (relu && !(a_y[k] > 0.f)) dz = 0.f;
    const float xh = (a_x[k] - mu) * rs;
    out[k] = rs * (ga * dz - fm1 - xh * fm2);
  }
  if (V == 4)
    *reinterpret_cast<float4*>(dx + idx) = make_float4(out[0], out[1], out[2], out[3]);
  else
    dx[idx] = out[0];
}

namespace old_r5 {
template <int V>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const float* __restrict__ dy, const float* __restrict__ y,
    const float* __restrict__ x, Levels lv, int C, int G,
    const float* __restrict__ mean, const float* __restrict__ rstd,
    const float* __restrict__ gamma, const float* __restrict__ gm, int relu,
    float* __restrict__ dx) {
  const int row = blockIdx.y;
  const int c = row % C, n = row / C;
  const int p = (blockIdx.x * 256 + threadIdx.x) * V;
  if (p >= lv.P) return;
  const int L = lv.num_levels, g = c / (C / G);
  const size_t ob = ((size_t)n * G + g) * L;
  const size_t idx = (size_t)row * lv.P + p;
  const float ga = gamma[c];
  float a_dy[V], a_y[V], a_x[V], out[V];
  if (V == 4) {
    const float4 t0 = *reinterpret_cast<const float4*>(dy + idx);
    const float4 t2 = *reinterpret_cast<const float4*>(x + idx);
    a_dy[0] = t0.x; a_dy[1] = t0.y; a_dy[2] = t0.z; a_dy[3] = t0.w;
    a_x[0] = t2.x; a_x[1] = t2.y; a_x[2] = t2.z; a_x[3] = t2.w;
    if (relu) {
      const float4 t1 = *reinterpret_cast<const float4*>(y + idx);
      a_y[0] = t1.x; a_y[1] = t1.y; a_y[2] = t1.z; a_y[3] = t1.w;
    }
  } else {
    a_dy[0] = dy[idx];
    a_x[0] = x[idx];
    if (relu) a_y[0] = y[idx];
  }
#pragma unroll
  for (int k = 0; k < V; ++k) {
    const int l = level_of_pos(lv, p + k);
    const float mu = mean[ob + l], rs = rstd[ob + l];
    const float fm1 = gm[(ob + l) * 2 + 0], fm2 = gm[(ob + l) * 2 + 1];
    float dz = a_dy[k];
    if (relu && !(a_y[k] > 0.f)) dz = 0.f;
    const float xh = (a_x[k] - mu) * rs;
    out[k] = rs * (ga * dz - fm1 - xh * fm2);
  }
  if (V == 4)
    *reinterpret_cast<float4*>(dx + idx) = make_float4(out[0], out[1], out[2], out[3]);
  else
    dx[idx] = out[0];
}

}  // namespace old_r5

// dgamma[c] = sum_{n,l} s2, dbeta[c] = sum_{n,l} s1
__global__ void gn_bwd_param_kernel(const double* __restrict__ sums, int N, int C,
                                    int L, float* __restrict__ dgamma,
                                    float* __restrict__ dbeta, int accumulate) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int n = 0; n < N; ++n)
    for (int l = 0; l < L; ++l) {
      const size_t r = ((size_t)(n * C + c) * L + l) * kGnBwdSplit * 2;
      s1 += sums[r + 0];
      s2 += sums[r + 1];
    }
  dbeta[c] = accumulate ? dbeta[c] + (float)s1 : (float)s1;
  dgamma[c] = accumulate ? dgamma[c] + (float)s2 : (float)s2;
}

// --------------------------------------------------------------- max pool ---
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(
    const float* __restrict__ x, int rows, int H, int W, int Ho, int Wo,
    float* __restrict__ y) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)rows * Ho * Wo;
  if (i >= total) return;
  const int wo = (int)(i % Wo);
  const size_t q = i / Wo;
  const int ho = (int)(q % Ho);
  const size_t r = q / Ho;
  const float* xp = x + r * H * W;
  float m = -INFINITY;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = ho * 2 - 1 + kh;
    if (hi < 0 || hi >= H) continue;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
      const int wi = wo * 2 - 1 + kw;
      if (wi < 0 || wi >= W) continue;
      m = fmaxf(m, xp[(size_t)hi * W + wi]);
    }
  }
  y[i] = m;
}

// Vector form for the stem (400 x 672 -> 200 x 336 x 64 channels x 2 images:
// 137 MB in, 34 MB out, purely HBM-bound; the scalar kernel above moved it at
// 1.8 TB/s with nine 4-byte loads at an 8-byte lane stride per output).  One
// thread = two adjacent outputs of a row: input columns 4t-1 .. 4t+3, i.e. ONE
// aligned 16-byte load per input row plus the previous thread's last column,
// which arrives by a lane shuffle (a 4-byte load only at a wave's first lane).
// Needs W % 4 == 0 (then Wo = W / 2 is even) and 16-byte aligned rows.
__global__ __launch_bounds__(256) void maxpool3x3s2_vec_kernel(
    const float* __restrict__ x, int rows, int H, int W, int Ho, int Wo,
    float* __restrict__ y) {
  const int half = Wo >> 1;                       // threads per output row
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t total = (size_t)rows * Ho * half;
  const bool live = i < total;
  const size_t ii = live ? i : total - 1;
  const int t = (int)(ii % half);
  const size_t q = ii / half;
  const int ho = (int)(q % Ho);
  const size_t r = q / Ho;
  const float* xp = x + r * H * W;
  float m0 = -INFINITY, m1 = -INFINITY;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int kh = 0; kh < 3; ++kh) {
    const int hi = ho * 2 - 1 + kh;
    const bool ok = hi >= 0 && hi < H;
    float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (ok) v = *reinterpret_cast<const float4*>(xp + (size_t)hi * W + 4 * t);
    // column 4t - 1 = the left neighbour's v.w when that neighbour is the
    // previous thread of the SAME output row (same wave, t > 0)
    float left = __shfl_up(v.w, 1);
    if (lane == 0 || t == 0)
      left = (ok && t > 0) ? xp[(size_t)hi * W + 4 * t - 1] : -INFINITY;
    m0 = fmaxf(m0, fmaxf(left, fmaxf(v.x, v.y)));
    m1 = fmaxf(m1, fmaxf(v.y, fmaxf(v.z, v.w)));
  }
  if (live) {
    float2 o = make_float2(m0, m1);
    *reinterpret_cast<float2*>(y + (r * Ho + ho) * Wo + 2 * t) = o;
  }
}

// ---------------------------------------------------- FPN top-down pathway --
// out = fine + nearest_up(coarse), target size = the finer map's size
// (fpn.py:182-191: laterals[i-1] += F.interpolate(laterals[i], size=prev)).
__device__ __forceinline__ int nearest_src(int dst, int in, int out) {
  // F.interpolate(mode='nearest'): min(floor(dst * in/out), in-1), fp32 scale
  const float scale = (float)in / (float)out;
  return min((int)floorf((float)dst * scale), in - 1);
}

__global__ __launch_bounds__(256) void upsample_add_fwd_kernel(
    const float* __restrict__ fine, const float* __restrict__ coarse, int Hf, int Wf,
    int Hc, int Wc, float* __restrict__ out) {
  const int row = blockIdx.y;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= Hf * Wf) return;
  const int h = r / Wf, w = r - h * Wf;
  const int hc = nearest_src(h, Hc, Hf), wc = nearest_src(w, Wc, Wf);
  out[(size_t)row * Hf * Wf + r] =
      fine[(size_t)row * Hf * Wf + r] + coarse[(size_t)row * Hc * Wc + hc * Wc + wc];
}

// dcoarse[q] = sum over fine cells mapping to q of dout
// addend (optional, round 5): the gradient the coarse map already received from
// its other consumer (the level's 3x3 output conv) -- summed here instead of by
// an elementwise launch of the autograd engine
__global__ __launch_bounds__(256) void upsample_add_bwd_kernel(
    const float* __restrict__ dout, int Hf, int Wf, int Hc, int Wc,
    const float* addend, float* dcoarse) {
  const int row = blockIdx.y;
  const int r = blockIdx.x * 256 + threadIdx.x;
  if (r >= Hc * Wc) return;
  const int h = r / Wc, w = r - h * Wc;
  const int h0 = max(0, (int)((long long)h * Hf / Hc) - 2);
  const int h1 = min(Hf - 1, (int)((long long)(h + 1) * Hf / Hc) + 2);
  const int w0 = max(0, (int)((long long)w * Wf / Wc) - 2);
  const int w1 = min(Wf - 1, (int)((long long)(w + 1) * Wf / Wc) + 2);
  float acc = 0.f;
  for (int hf = h0; hf <= h1; ++hf) {
    if (nearest_src(hf, Hc, Hf) != h) continue;
    for (int wf = w0; wf <= w1; ++wf) {
      if (nearest_src(wf, Wc, Wf) != w) continue;
      acc += dout[(size_t)row * Hf * Wf + hf * Wf + wf];
    }
  }
  const size_t o = (size_t)row * Hc * Wc + r;
  dcoarse[o] = addend ? acc + addend[o] : acc;
}

// (N*C, P) level-concatenated rows <-> per-level contiguous (N*C, H_l*W_l)
// tensors, all levels in one launch (pack = what torch.cat(dim=2) did for the head
// input, gfl_head.py:164-172 runs the shared towers on every level; unpack = its
// backward).  16-byte accesses where the level length allows.
struct LevelPtrs {
  float* p[LD_MAX_LEVELS];
};
template <bool PACK>
__global__ __launch_bounds__(256) void pack_levels_kernel(Levels lv, LevelPtrs lp,
                                                          float* __restrict__ x3) {
  const int row = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= lv.P) return;
  const int l = level_of_pos(lv, p);
  const int len = lv.off[l + 1] - lv.off[l];
  float* a = lp.p[l] + (size_t)row * len + (p - lv.off[l]);
  float* b = x3 + (size_t)row * lv.P + p;
  if (PACK) *b = *a;
  else *a = *b;
}

// The same with the bf16 C8 image of the destination as a side output (bf16 mode:
// the packed head input feeds the first tower convs, the unpacked level gradients
// feed the neck convs' data / weight gradients -- no conversion launch): thread =
// (n, 8 channels, one position); c8_levels[l] / x3_c8 = (N, C/8, len, 8) images.
typedef float pk_floatx8 __attribute__((ext_vector_type(8)));
typedef __bf16 pk_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned pk_uintx4 __attribute__((ext_vector_type(4)));
struct LevelC8Ptrs {
  pk_uintx4* p[LD_MAX_LEVELS];
};
template <bool PACK>
__global__ __launch_bounds__(256) void pack_levels_c8_kernel(Levels lv, LevelPtrs lp, int C,
                                                             float* __restrict__ x3,
                                                             LevelC8Ptrs c8l,
                                                             pk_uintx4* __restrict__ x3_c8) {
  const int C8 = C >> 3;
  const int blk = blockIdx.y;  // n * C8 + c8
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= lv.P) return;
  const int l = level_of_pos(lv, p);
  const int len = lv.off[l + 1] - lv.off[l], q = p - lv.off[l];
  pk_floatx8 v;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const size_t row = (size_t)blk * 8 + e;  // n * C + c
    float* a = lp.p[l] + row * len + q;
    float* b = x3 + row * lv.P + p;
    if (PACK) {
      v[e] = *a;
      *b = v[e];
    } else {
      v[e] = *b;
      *a = v[e];
    }
  }
  const pk_uintx4 w = __builtin_bit_cast(pk_uintx4, __builtin_convertvector(v, pk_bf16x8));
  if (PACK) x3_c8[(size_t)blk * lv.P + p] = w;
  else c8l.p[l][(size_t)blk * len + q] = w;
}

// ------------------------------------------------------------- Scale layer --
// y[n,c,p] = x[n,c,p] * scale[level(p)]
__global__ __launch_bounds__(256) void scale_levels_kernel(
    const float* __restrict__ x, Levels lv, const float* __restrict__ scales,
    float* __restrict__ y) {
  const int row = blockIdx.y;
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= lv.P) return;
  const int l = level_of_pos(lv, p);
  const size_t idx = (size_t)row * lv.P + p;
  y[idx] = x[idx] * scales[l];
}

// dscale[l] = sum dy * x over level l (all rows): kScaleSplit blocks per level
// write fp64 partials, a second tiny kernel adds them in fixed order
constexpr int kScaleSplit = 64;

__global__ __launch_bounds__(256) void scale_levels_bwd_partial_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, Levels lv, int rows,
    double* __restrict__ partial) {
  const int l = blockIdx.x / kScaleSplit, sp = blockIdx.x % kScaleSplit;
  const int A = lv.off[l + 1] - lv.off[l];
  const long long total = (long long)rows * A;
  const long long per = (total + kScaleSplit - 1) / kScaleSplit;
  const long long beg = sp * per, end = min(total, beg + per);
  double s = 0.0, z = 0.0;
  for (long long e = beg + threadIdx.x; e < end; e += 256) {
    const int r = (int)(e / A), p = (int)(e - (long long)r * A);
    const size_t idx = (size_t)r * lv.P + lv.off[l] + p;
    s += (double)(dy[idx] * x[idx]);
  }
  block_sum2(s, z);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void scale_levels_bwd_final_kernel(const double* __restrict__ partial,
                                              int L, float* __restrict__ dscale,
                                              int accumulate) {
  const int l = threadIdx.x;
  if (l >= L) return;
  double s = 0.0;
  for (int k = 0; k < kScaleSplit; ++k) s += partial[l * kScaleSplit + k];
  dscale[l] = accumulate ? dscale[l] + (float)s : (float)s;
}

// --------------------------------------------------------------------- SGD --
// torch.optim.SGD: d = g + wd*p; buf = mu*buf + d; p -= lr*buf
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p,
                                                  const float* __restrict__ g,
                                                  float* __restrict__ buf, size_t n,
                                                  float lr, float mu, float wd,
                                                  float gscale) {
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 pv = *reinterpret_cast<float4*>(p + i);
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    float4 bv = *reinterpret_cast<float4*>(buf + i);
    bv.x = mu * bv.x + (gv.x * gscale + wd * pv.x);
    bv.y = mu * bv.y + (gv.y * gscale + wd * pv.y);
    bv.z = mu * bv.z + (gv.z * gscale + wd * pv.z);
    bv.w = mu * bv.w + (gv.w * gscale + wd * pv.w);
    pv.x -= lr * bv.x; pv.y -= lr * bv.y; pv.z -= lr * bv.z; pv.w -= lr * bv.w;
    *reinterpret_cast<float4*>(p + i) = pv;
    *reinterpret_cast<float4*>(buf + i) = bv;
  } else {
    for (size_t k = i; k < n; ++k) {
      const float b = mu * buf[k] + (g[k] * gscale + wd * p[k]);
      buf[k] = b;
      p[k] -= lr * b;
    }
  }
}

// the same update with (lr, mu, wd, gscale) read from device memory: a captured
// hipGraph replays the launch with its ARGUMENTS frozen, so a learning-rate
// schedule must reach the kernel through a buffer the host rewrites between
// replays (train.GraphedStep)
__global__ __launch_bounds__(256) void sgd_dev_kernel(float* __restrict__ p,
                                                      const float* __restrict__ g,
                                                      float* __restrict__ buf, size_t n,
                                                      const float* __restrict__ hyper) {
  const float lr = hyper[0], mu = hyper[1], wd = hyper[2], gscale = hyper[3];
  const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i + 3 < n) {
    float4 pv = *reinterpret_cast<float4*>(p + i);
    const float4 gv = *reinterpret_cast<const float4*>(g + i);
    float4 bv = *reinterpret_cast<float4*>(buf + i);
    bv.x = mu * bv.x + (gv.x * gscale + wd * pv.x);
    bv.y = mu * bv.y + (gv.y * gscale + wd * pv.y);
    bv.z = mu * bv.z + (gv.z * gscale + wd * pv.z);
    bv.w = mu * bv.w + (gv.w * gscale + wd * pv.w);
    pv.x -= lr * bv.x; pv.y -= lr * bv.y; pv.z -= lr * bv.z; pv.w -= lr * bv.w;
    *reinterpret_cast<float4*>(p + i) = pv;
    *reinterpret_cast<float4*>(buf + i) = bv;
  } else {
    for (size_t k = i; k < n; ++k) {
      const float b = mu * buf[k] + (g[k] * gscale + wd * p[k]);
      buf[k] = b;
      p[k] -= lr * b;
    }
  }
}

Levels make_levels(const ld_levels_t* lv) {
  Levels k;
  k.num_levels = lv->num_levels;
  int off = 0;
  for (int l = 0; l < LD_MAX_LEVELS + 1; ++l) k.off[l] = 0;
  for (int l = 0; l < lv->num_levels; ++l) {
    k.off[l] = off;
    off += lv->H[l] * lv->W[l];
  }
  for (int l = lv->num_levels; l < LD_MAX_LEVELS + 1; ++l) k.off[l] = off;
  k.P = off;
  return k;
}

int check_levels(const ld_levels_t* lv) {
  if (!lv || lv->num_levels < 1 || lv->num_levels > LD_MAX_LEVELS) return LD_EINVAL;
  for (int l = 0; l < lv->num_levels; ++l)
    if (lv->H[l] < 1 || lv->W[l] < 1) return LD_EINVAL;
  return 0;
}

constexpr int kBnSplitMax = 256;

int bn_splits(int N, int C, int P) {
  const long long per_c = (long long)N * P;
  int s = (int)((2048 + C - 1) / C);  // aim at ~2048 blocks
  const int max_by_work = (int)((per_c + 4095) / 4096);
  if (s > max_by_work) s = max_by_work;
  if (s < 1) s = 1;
  if (s > kBnSplitMax) s = kBnSplitMax;
  return s;
}

}  // namespace

#define LD_STREAM ((hipStream_t)stream)

extern "C" int ld_bn_prepare(const float* gamma, const float* beta,
                             const float* mean, const float* var, float eps, int C,
                             float* scale, float* shift, float* rstd,
                             ld_stream_t stream) {
  if (!gamma || !beta || !mean || !var || !scale || !shift || C < 1)
    return LD_EINVAL;
  LD_LAUNCH(bn_prepare_kernel, dim3((C + 255) / 256), dim3(256), 0,
                     LD_STREAM, gamma, beta, mean, var, eps, C, scale, shift, rstd);
  return (int)hipGetLastError();
}

__global__ __launch_bounds__(256) void bn_prepare_batch_kernel(
    const ld_bn_job_t* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  const ld_bn_job_t j = jobs[block_job[blockIdx.x]];
  const int c = (blockIdx.x - j.first_block) * 256 + threadIdx.x;
  if (c >= j.C) return;
  const float r = 1.0f / sqrtf(j.var[c] + j.eps);
  const float s = j.gamma[c] * r;
  j.scale[c] = s;
  j.shift[c] = j.beta[c] - j.mean[c] * s;
  if (j.rstd) j.rstd[c] = r;
}

extern "C" int ld_bn_prepare_batch(const ld_bn_job_t* jobs, const int32_t* block_job,
                                   int nblocks, ld_stream_t stream) {
  if (!jobs || !block_job || nblocks < 1) return LD_EINVAL;
  LD_LAUNCH(bn_prepare_batch_kernel, dim3(nblocks), dim3(256), 0, LD_STREAM,
                     jobs, block_job);
  return (int)hipGetLastError();
}

extern "C" int ld_bn_act_forward_c8(const float* x, const float* residual,
                                    const float* scale, const float* shift, int N,
                                    int C, int P, int relu, float* y, void* y_c8,
                                    ld_stream_t stream) {
  if (!x || !scale || !shift || !y || !y_c8 || N < 1 || C < 1 || P < 1)
    return LD_EINVAL;
  if (P % 4 != 0 || C % 8 != 0 || (uintptr_t)x % 16 != 0 || (uintptr_t)y % 16 != 0 ||
      (residual && (uintptr_t)residual % 16 != 0))
    return LD_EUNSUPPORTED;
  LD_LAUNCH(bn_act_fwd_c8_kernel, dim3((P / 4 + 255) / 256, N * (C / 8)),
                     dim3(256), 0, LD_STREAM, x, residual, scale, shift, C, P, relu, y,
                     (gn_uintx4*)y_c8);
  return (int)hipGetLastError();
}

extern "C" int ld_bn_act_forward(const float* x, const float* residual,
                                 const float* scale, const float* shift, int N,
                                 int C, int P, int relu, float* y,
                                 ld_stream_t stream) {
  if (!x || !scale || !shift || !y || N < 1 || C < 1 || P < 1) return LD_EINVAL;
  const bool vec = (P % 4 == 0) && ((uintptr_t)x % 16 == 0) &&
                   ((uintptr_t)y % 16 == 0) &&
                   (!residual || (uintptr_t)residual % 16 == 0);
  if (vec)
    LD_LAUNCH((bn_act_fwd_kernel<true>), dim3((P / 4 + 255) / 256, N * C),
                       dim3(256), 0, LD_STREAM, x, residual, scale, shift, C, P,
                       relu, y);
  else
    LD_LAUNCH((bn_act_fwd_kernel<false>), dim3((P + 255) / 256, N * C),
                       dim3(256), 0, LD_STREAM, x, residual, scale, shift, C, P,
                       relu, y);
  return (int)hipGetLastError();
}

extern "C" size_t ld_bn_act_backward_workspace_bytes(int N, int C, int P) {
  if (N < 1 || C < 1 || P < 1) return 0;
  return (size_t)C * kBnSplitMax * 2 * sizeof(double);
}

extern "C" int ld_bn_act_backward(const float* dy, const float* y, const float* x,
                                  const float* scale, const float* mean,
                                  const float* rstd, int N, int C, int P, int relu,
                                  float* dx, float* dres, float* dgamma,
                                  float* dbeta, int accumulate, void* workspace,
                                  size_t workspace_bytes, ld_stream_t stream) {
  if (!dy || !scale || N < 1 || C < 1 || P < 1) return LD_EINVAL;
  if (relu && !y) return LD_EINVAL;
  const bool params = dgamma || dbeta;
  if (params && (!x || !mean || !rstd)) return LD_EINVAL;
  if (params && (!workspace ||
                 workspace_bytes < ld_bn_act_backward_workspace_bytes(N, C, P)))
    return LD_ENOSPACE;
  const int ns = bn_splits(N, C, P);
  const bool vec = P % 4 == 0 &&
                   ((uintptr_t)dy | (uintptr_t)(y ? y : dy) | (uintptr_t)(x ? x : dy) |
                    (uintptr_t)(dx ? dx : dy) | (uintptr_t)(dres ? dres : dy)) % 16 == 0;
  if (vec)
    LD_LAUNCH((bn_act_bwd_kernel<true>), dim3(C, ns), dim3(256), 0, LD_STREAM,
                       dy, y, x, scale, mean, rstd, N, C, P, relu, dx, dres,
                       params ? (double*)workspace : nullptr);
  else
    LD_LAUNCH((bn_act_bwd_kernel<false>), dim3(C, ns), dim3(256), 0,
                       LD_STREAM, dy, y, x, scale, mean, rstd, N, C, P, relu, dx, dres,
                       params ? (double*)workspace : nullptr);
  if (params && accumulate != LD_GRAD_DEFER)
    LD_LAUNCH(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0,
                       LD_STREAM, (const double*)workspace, C, ns, dgamma, dbeta,
                       accumulate);
  return (int)hipGetLastError();
}

extern "C" int ld_bn_act_backward_c8(const float* dy, const float* y, const float* x,
                                     const float* scale, const float* mean,
                                     const float* rstd, int N, int C, int P, int relu,
                                     float* dx, void* dx_c8, float* dres, float* dgamma,
                                     float* dbeta, int accumulate, void* workspace,
                                     size_t workspace_bytes, ld_stream_t stream) {
  if (!dy || !scale || !dx_c8 || N < 1 || C < 1 || P < 1) return LD_EINVAL;
  if (relu && !y) return LD_EINVAL;
  const bool params = dgamma || dbeta;
  if (params && (!x || !mean || !rstd)) return LD_EINVAL;
  if (params && (!workspace ||
                 workspace_bytes < ld_bn_act_backward_workspace_bytes(N, C, P)))
    return LD_ENOSPACE;
  const int xb = (P / 4 + 63) / 64, nslots = N * xb;
  if (P % 4 != 0 || C % 8 != 0 || nslots > kBnSplitMax ||
      ((uintptr_t)dy | (uintptr_t)(y ? y : dy) | (uintptr_t)(x ? x : dy) |
       (uintptr_t)(dx ? dx : dy) | (uintptr_t)(dres ? dres : dy) | (uintptr_t)dx_c8) % 16)
    return LD_EUNSUPPORTED;
  LD_LAUNCH(bn_act_bwd_c8_kernel, dim3(xb, N * (C / 8)), dim3(64), 0, LD_STREAM,
                     dy, y, x, scale, mean, rstd, C, P, relu, dx, dres,
                     (gn_uintx4*)dx_c8, params ? (double*)workspace : nullptr, nslots);
  if (params && accumulate != LD_GRAD_DEFER)
    LD_LAUNCH(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0,
                       LD_STREAM, (const double*)workspace, C, nslots, dgamma, dbeta,
                       accumulate);
  return (int)hipGetLastError();
}

// ld_bn_act_backward_c8 with y and x given as bf16 C8 images (N, C/8, P, 8) instead
// of fp32 tensors (round 6; y_c8 may be NULL without relu).  Same outputs, same
// partial-slot layout (ld_bn_act_backward_nsplit(..., c8 = 1)); dgamma is computed
// from the bf16-rounded conv result.
extern "C" int ld_bn_act_backward_c8in(const float* dy, const void* y_c8,
                                       const void* x_c8, const float* scale,
                                       const float* mean, const float* rstd, int N,
                                       int C, int P, int relu, float* dx, void* dx_c8,
                                       float* dres, float* dgamma, float* dbeta,
                                       int accumulate, void* workspace,
                                       size_t workspace_bytes, ld_stream_t stream) {
  if (!dy || !scale || !dx_c8 || !x_c8 || N < 1 || C < 1 || P < 1) return LD_EINVAL;
  if (relu && !y_c8) return LD_EINVAL;
  const bool params = dgamma || dbeta;
  if (params && (!mean || !rstd)) return LD_EINVAL;
  if (params && (!workspace ||
                 workspace_bytes < ld_bn_act_backward_workspace_bytes(N, C, P)))
    return LD_ENOSPACE;
  // 4 positions per thread where the fp32 rows are 16-byte aligned (P % 4 == 0), 2
  // where they are 8-byte aligned (P % 2 == 0: the 25 x 42 stage, P = 1050)
  const int V = P % 4 == 0 ? 4 : 2;
  const int xb = (P / V + 63) / 64, nslots = N * xb;
  if (P % 2 != 0 || C % 8 != 0 || nslots > kBnSplitMax ||
      ((uintptr_t)(y_c8 ? y_c8 : x_c8) | (uintptr_t)x_c8 | (uintptr_t)dx_c8) % 16 ||
      ((uintptr_t)dy | (uintptr_t)(dx ? dx : dy) | (uintptr_t)(dres ? dres : dy)) %
          (4 * V))
    return LD_EUNSUPPORTED;
  if (V == 4)
    LD_LAUNCH(bn_act_bwd_c8in_kernel<4>, dim3(xb, N * (C / 8)), dim3(64), 0, LD_STREAM,
                       dy, (const gn_uintx4*)y_c8, (const gn_uintx4*)x_c8, scale, mean,
                       rstd, C, P, relu, dx, dres, (gn_uintx4*)dx_c8,
                       params ? (double*)workspace : nullptr, nslots);
  else
    LD_LAUNCH(bn_act_bwd_c8in_kernel<2>, dim3(xb, N * (C / 8)), dim3(64), 0, LD_STREAM,
                       dy, (const gn_uintx4*)y_c8, (const gn_uintx4*)x_c8, scale, mean,
                       rstd, C, P, relu, dx, dres, (gn_uintx4*)dx_c8,
                       params ? (double*)workspace : nullptr, nslots);
  if (params && accumulate != LD_GRAD_DEFER)
    LD_LAUNCH(bn_bwd_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0,
                       LD_STREAM, (const double*)workspace, C, nslots, dgamma, dbeta,
                       accumulate);
  return (int)hipGetLastError();
}

// How many partial slots per channel ld_bn_act_backward (c8 == 0) /
// ld_bn_act_backward_c8 (c8 != 0) write for this geometry (host logic).
extern "C" int ld_bn_act_backward_nsplit(int N, int C, int P, int c8) {
  if (N < 1 || C < 1 || P < 1) return 0;
  // c8 == 1: ld_bn_act_backward_c8 (P % 4 == 0); c8 == 2: ld_bn_act_backward_c8in
  // (4 positions per thread, or 2 when P % 4 != 0)
  if (c8 == 2) return N * ((P / (P % 4 == 0 ? 4 : 2) + 63) / 64);
  return c8 ? N * ((P / 4 + 63) / 64) : bn_splits(N, C, P);
}

// every job of the table in ONE launch: block b finalises 16 channels of job
// block_job[b] with bn_bwd_finalize_kernel's own arithmetic (16 lanes per channel
// over the slots, fixed butterfly)
__global__ __launch_bounds__(256) void bn_bwd_finalize_batch_kernel(
    const ld_bn_fin_job_t* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  const ld_bn_fin_job_t jb = jobs[block_job[blockIdx.x]];
  const int j = threadIdx.x & 15;
  const int c = (blockIdx.x - jb.first_block) * 16 + (threadIdx.x >> 4);
  double s1 = 0.0, s2 = 0.0;
  if (c < jb.C) {
    const double2* row = reinterpret_cast<const double2*>(jb.partial) + (size_t)c * jb.nsplit;
    for (int k = j; k < jb.nsplit; k += 16) {
      const double2 v = row[k];
      s1 += v.x;
      s2 += v.y;
    }
  }
#pragma unroll
  for (int m = 8; m >= 1; m >>= 1) {
    s1 += __shfl_xor(s1, m, 16);
    s2 += __shfl_xor(s2, m, 16);
  }
  if (c >= jb.C || j != 0) return;
  if (jb.dbeta) jb.dbeta[c] = jb.accumulate ? jb.dbeta[c] + (float)s1 : (float)s1;
  if (jb.dgamma) jb.dgamma[c] = jb.accumulate ? jb.dgamma[c] + (float)s2 : (float)s2;
}

extern "C" int ld_bn_bwd_finalize_batch(const ld_bn_fin_job_t* jobs,
                                        const int32_t* block_job, int nblocks,
                                        ld_stream_t stream) {
  if (!jobs || !block_job || nblocks < 1) return LD_EINVAL;
  LD_LAUNCH(bn_bwd_finalize_batch_kernel, dim3(nblocks), dim3(256), 0, LD_STREAM,
                     jobs, block_job);
  return (int)hipGetLastError();
}

extern "C" int ld_bias_grad_nsplit(int N, int C, int P) {
  if (N < 1 || C < 1 || P < 1) return 0;
  return bn_splits(N, C, P);
}

// partial must hold C * ld_bias_grad_nsplit(N, C, P) * 2 doubles
extern "C" int ld_bias_grad_partial(const float* dy, int N, int C, int P, void* partial,
                                    size_t partial_bytes, ld_stream_t stream) {
  if (!dy || !partial || N < 1 || C < 1 || P < 1) return LD_EINVAL;
  const int ns = bn_splits(N, C, P);
  if (partial_bytes < (size_t)C * ns * 2 * sizeof(double)) return LD_ENOSPACE;
  LD_LAUNCH(bias_grad_partial_kernel, dim3(C, ns), dim3(256), 0, LD_STREAM, dy, N, C, P, ns,
            (double*)partial);
  return (int)hipGetLastError();
}

extern "C" int ld_bias_grad(const float* dy, int N, int C, int P, float* db,
                            int accumulate, ld_stream_t stream) {
  if (!dy || !db || N < 1 || C < 1 || P < 1) return LD_EINVAL;
  LD_LAUNCH(bias_grad_kernel, dim3(C), dim3(256), 0, LD_STREAM, dy, N, C, P,
                     db, accumulate);
  return (int)hipGetLastError();
}

extern "C" size_t ld_gn_forward_workspace_bytes(const ld_levels_t* lv, int N,
                                                int G) {
  if (check_levels(lv) != 0 || N < 1 || G < 1) return 0;
  return (size_t)N * G * lv->num_levels * kGnSplit * 2 * sizeof(double);
}

static int gn_forward_impl(const ld_levels_t* lv, const float* x, const float* gamma,
                           const float* beta, int N, int C, int G, float eps,
                           int relu, float* y, void* y_c8, float* mean, float* rstd,
                           void* workspace, size_t workspace_bytes,
                           ld_stream_t stream) {
  if (int e = check_levels(lv)) return e;
  if (!x || !gamma || !beta || (!y && !y_c8) || !mean || !rstd || N < 1 || C < 1 ||
      G < 1 || C % G)
    return LD_EINVAL;
  if (!workspace || workspace_bytes < ld_gn_forward_workspace_bytes(lv, N, G))
    return LD_ENOSPACE;
  const Levels k = make_levels(lv);
  const int ngl = N * G * k.num_levels;
  int per_ng = 0;
  for (int l = 0; l < k.num_levels; ++l) per_ng += gn_slices(k.off[l + 1] - k.off[l]);
  LD_LAUNCH(gn_stats_partial_kernel, dim3(N * G * per_ng), dim3(256), 0,
                     LD_STREAM, x, k, C, G, (double*)workspace);
  LD_LAUNCH(gn_stats_final_kernel, dim3((ngl + 255) / 256), dim3(256), 0,
                     LD_STREAM, (const double*)workspace, k, N, C, G, eps, mean,
                     rstd);
  if (y_c8) {
    if (k.P % 4 != 0 || C % 8 != 0 ||
        ((uintptr_t)x | (uintptr_t)y | (uintptr_t)y_c8) % 16 != 0)
      return LD_EUNSUPPORTED;
    LD_LAUNCH(gn_apply_c8_kernel, dim3((k.P / 4 + 255) / 256, N * (C / 8)),
                       dim3(256), 0, LD_STREAM, x, k, C, G, mean, rstd, gamma, beta,
                       relu, y, (gn_uintx4*)y_c8);
  } else if (k.P % 4 == 0 && ((uintptr_t)x | (uintptr_t)y) % 16 == 0)
    LD_LAUNCH(gn_apply4_kernel, dim3((k.P / 4 + 255) / 256, N * C),
                       dim3(256), 0, LD_STREAM, x, k, C, G, mean, rstd, gamma, beta,
                       relu, y);
  else
    LD_LAUNCH(gn_apply_kernel, dim3((k.P + 255) / 256, N * C), dim3(256), 0,
                       LD_STREAM, x, k, C, G, mean, rstd, gamma, beta, relu, y);
  return (int)hipGetLastError();
}

extern "C" int ld_gn_forward(const ld_levels_t* lv, const float* x,
                             const float* gamma, const float* beta, int N, int C,
                             int G, float eps, int relu, float* y, float* mean,
                             float* rstd, void* workspace, size_t workspace_bytes,
                             ld_stream_t stream) {
  return gn_forward_impl(lv, x, gamma, beta, N, C, G, eps, relu, y, nullptr, mean,
                         rstd, workspace, workspace_bytes, stream);
}

extern "C" int ld_gn_forward_c8(const ld_levels_t* lv, const float* x,
                                const float* gamma, const float* beta, int N, int C,
                                int G, float eps, int relu, float* y, void* y_c8,
                                float* mean, float* rstd, void* workspace,
                                size_t workspace_bytes, ld_stream_t stream) {
  if (!y_c8) return LD_EINVAL;
  return gn_forward_impl(lv, x, gamma, beta, N, C, G, eps, relu, y, y_c8, mean, rstd,
                         workspace, workspace_bytes, stream);
}

static size_t gn_bwd_sums_bytes(int L, int N, int C) {
  return (size_t)N * C * L * kGnBwdSplit * 2 * sizeof(double);
}

extern "C" size_t ld_gn_backward_workspace_bytes(const ld_levels_t* lv, int N,
                                                 int C) {
  if (check_levels(lv) != 0 || N < 1 || C < 1) return 0;
  // slice sums (fp64) + per (n, group, level) means (fp32; G <= C)
  return gn_bwd_sums_bytes(lv->num_levels, N, C) +
         (size_t)N * C * lv->num_levels * 2 * sizeof(float);
}

// beta != null = the lean form (ld_gn_backward_c8_lean): the ReLU mask is recomputed
// from x, y is not read (may be null), dx may be null (dx_c8 is then the only output)
static int gn_backward_impl(const ld_levels_t* lv, const float* dy, const float* y,
                            const float* x, const float* gamma, const float* beta,
                            const float* mean,
                            const float* rstd, int N, int C, int G, int relu,
                            float* dx, void* dx_c8, float* dgamma, float* dbeta,
                            int accumulate, void* workspace, size_t workspace_bytes,
                            ld_stream_t stream) {
  if (int e = check_levels(lv)) return e;
  const bool lean = beta != nullptr;
  if (!dy || !x || !gamma || !mean || !rstd || (!dx && !(lean && dx_c8)) || N < 1 ||
      C < 1 || G < 1 || C % G)
    return LD_EINVAL;
  if (relu && !y && !lean) return LD_EINVAL;
  if (lean) y = dy;  // never dereferenced; keeps the alignment tests below uniform
  if (!workspace || workspace_bytes < ld_gn_backward_workspace_bytes(lv, N, C))
    return LD_ENOSPACE;
  const Levels k = make_levels(lv);
  double* sums = (double*)workspace;
  float* gm = (float*)((char*)workspace + gn_bwd_sums_bytes(k.num_levels, N, C));
  const int rl = N * C * k.num_levels, ngl = N * G * k.num_levels;
  const bool rowwise = k.P % 4 == 0 &&
                       ((uintptr_t)dy | (uintptr_t)x | (uintptr_t)(relu ? y : x)) % 16 == 0;
  if (lean && !rowwise) return LD_EUNSUPPORTED;
  if (rowwise && lean) {
    LD_LAUNCH(gn_bwd_reduce_row_kernel<true>, dim3(N * C), dim3(256), 0, LD_STREAM,
                       dy, y, x, k, C, G, mean, rstd, gamma, beta, relu, sums);
  } else if (rowwise) {
    LD_LAUNCH(gn_bwd_reduce_row_kernel<false>, dim3(N * C), dim3(256), 0, LD_STREAM,
                       dy, y, x, k, C, G, mean, rstd, gamma, beta, relu, sums);
  } else {
    LD_LAUNCH(gn_bwd_reduce_kernel, dim3(rl * kGnBwdSplit), dim3(256), 0,
                       LD_STREAM, dy, y, x, k, C, G, mean, rstd, relu, sums);
    LD_LAUNCH(gn_bwd_fold_kernel, dim3((rl + 255) / 256), dim3(256), 0,
                       LD_STREAM, sums, rl);
  }
  LD_LAUNCH(gn_bwd_group_kernel, dim3((ngl + 255) / 256), dim3(256), 0,
                     LD_STREAM, sums, k, N, C, G, gamma, gm);
  const bool vec = k.P % 4 == 0 &&
                   ((uintptr_t)dy | (uintptr_t)x | (uintptr_t)(dx ? dx : x) |
                    (uintptr_t)(relu ? y : x)) % 16 == 0;
  const char* old_env = getenv("LD_NN_OLD");  // A/B: the round-5 kernels
  const bool use_old = old_env && old_env[0] == '1';
  if (lean) {
    if (!dx_c8 || !vec || C % 8 != 0) return LD_EUNSUPPORTED;
    LD_LAUNCH(gn_bwd_apply_c8_kernel<true>,
                       dim3((k.P / 4 + 255) / 256, N * (C / 8)), dim3(256), 0,
                       LD_STREAM, dy, y, x, k, C, G, mean, rstd, gamma, beta, gm, relu,
                       dx, (gn_uintx4*)dx_c8);
  } else if (dx_c8 && use_old) {
    if (!vec || C % 8 != 0) return LD_EUNSUPPORTED;
    LD_LAUNCH(old_r5::gn_bwd_apply_c8_kernel,
                       dim3((k.P / 4 + 255) / 256, N * (C / 8)), dim3(256), 0,
                       LD_STREAM, dy, y, x, k, C, G, mean, rstd, gamma, gm, relu, dx,
                       (gn_uintx4*)dx_c8);
  } else if (use_old && vec) {
    LD_LAUNCH(old_r5::gn_bwd_apply_kernel<4>, dim3((k.P / 4 + 255) / 256, N * C),
                       dim3(256), 0, LD_STREAM, dy, y, x, k, C, G, mean, rstd, gamma,
                       gm, relu, dx);
  } else if (dx_c8) {
    if (!vec || C % 8 != 0) return LD_EUNSUPPORTED;
    LD_LAUNCH(gn_bwd_apply_c8_kernel<false>,
                       dim3((k.P / 4 + 255) / 256, N * (C / 8)), dim3(256), 0,
                       LD_STREAM, dy, y, x, k, C, G, mean, rstd, gamma, beta, gm, relu,
                       dx, (gn_uintx4*)dx_c8);
  } else if (vec)
    LD_LAUNCH(gn_bwd_apply_kernel<4>, dim3((k.P / 4 + 255) / 256, N * C),
                       dim3(256), 0, LD_STREAM, dy, y, x, k, C, G, mean, rstd, gamma,
                       gm, relu, dx);
  else
    LD_LAUNCH(gn_bwd_apply_kernel<1>, dim3((k.P + 255) / 256, N * C),
                       dim3(256), 0, LD_STREAM, dy, y, x, k, C, G, mean, rstd, gamma,
                       gm, relu, dx);
  if (dgamma && dbeta)
    LD_LAUNCH(gn_bwd_param_kernel, dim3((C + 255) / 256), dim3(256), 0,
                       LD_STREAM, sums, N, C, k.num_levels, dgamma, dbeta,
                       accumulate);
  return (int)hipGetLastError();
}

extern "C" int ld_gn_backward(const ld_levels_t* lv, const float* dy, const float* y,
                              const float* x, const float* gamma, const float* mean,
                              const float* rstd, int N, int C, int G, int relu,
                              float* dx, float* dgamma, float* dbeta, int accumulate,
                              void* workspace, size_t workspace_bytes,
                              ld_stream_t stream) {
  return gn_backward_impl(lv, dy, y, x, gamma, nullptr, mean, rstd, N, C, G, relu, dx,
                          nullptr, dgamma, dbeta, accumulate, workspace,
                          workspace_bytes, stream);
}

extern "C" int ld_gn_backward_c8(const ld_levels_t* lv, const float* dy,
                                 const float* y, const float* x, const float* gamma,
                                 const float* mean, const float* rstd, int N, int C,
                                 int G, int relu, float* dx, void* dx_c8,
                                 float* dgamma, float* dbeta, int accumulate,
                                 void* workspace, size_t workspace_bytes,
                                 ld_stream_t stream) {
  if (!dx_c8) return LD_EINVAL;
  return gn_backward_impl(lv, dy, y, x, gamma, nullptr, mean, rstd, N, C, G, relu, dx,
                          dx_c8, dgamma, dbeta, accumulate, workspace, workspace_bytes,
                          stream);
}

// The lean GroupNorm (+ ReLU) backward of bf16 mode (round 6): y is not an operand
// -- the ReLU mask is recomputed from x, mean, rstd, gamma, beta with the forward's
// own expression (bit-identical to reading y back: tests/test_gpu_layers.py) -- and
// dx (fp32) may be NULL when every consumer takes the C8 image dx_c8.  18 instead of
// 30 bytes of HBM traffic per element over the two passes.
extern "C" int ld_gn_backward_c8_lean(const ld_levels_t* lv, const float* dy,
                                      const float* x, const float* gamma,
                                      const float* beta, const float* mean,
                                      const float* rstd, int N, int C, int G, int relu,
                                      float* dx, void* dx_c8, float* dgamma,
                                      float* dbeta, int accumulate, void* workspace,
                                      size_t workspace_bytes, ld_stream_t stream) {
  if (!dx_c8 || !beta) return LD_EINVAL;
  return gn_backward_impl(lv, dy, nullptr, x, gamma, beta, mean, rstd, N, C, G, relu,
                          dx, dx_c8, dgamma, dbeta, accumulate, workspace,
                          workspace_bytes, stream);
}

extern "C" int ld_maxpool3x3s2(const float* x, int rows, int H, int W, float* y,
                               ld_stream_t stream) {
  if (!x || !y || rows < 1 || H < 1 || W < 1) return LD_EINVAL;
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)rows * Ho * Wo;
  if (W % 4 == 0 && W >= 8 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 8) == 0) {
    const size_t threads = total / 2;  // Wo = W / 2 is even
    LD_LAUNCH(maxpool3x3s2_vec_kernel, dim3((unsigned)((threads + 255) / 256)),
                       dim3(256), 0, LD_STREAM, x, rows, H, W, Ho, Wo, y);
    return (int)hipGetLastError();
  }
  LD_LAUNCH(maxpool3x3s2_kernel, dim3((unsigned)((total + 255) / 256)),
                     dim3(256), 0, LD_STREAM, x, rows, H, W, Ho, Wo, y);
  return (int)hipGetLastError();
}

extern "C" int ld_upsample_add_forward(const float* fine, const float* coarse,
                                       int rows, int Hf, int Wf, int Hc, int Wc,
                                       float* out, ld_stream_t stream) {
  if (!fine || !coarse || !out || rows < 1 || Hf < 1 || Wf < 1 || Hc < 1 || Wc < 1)
    return LD_EINVAL;
  LD_LAUNCH(upsample_add_fwd_kernel, dim3((Hf * Wf + 255) / 256, rows),
                     dim3(256), 0, LD_STREAM, fine, coarse, Hf, Wf, Hc, Wc, out);
  return (int)hipGetLastError();
}

extern "C" int ld_upsample_add_backward_acc(const float* dout, int rows, int Hf, int Wf,
                                            int Hc, int Wc, const float* addend,
                                            float* dcoarse, ld_stream_t stream) {
  if (!dout || !dcoarse || rows < 1 || Hf < 1 || Wf < 1 || Hc < 1 || Wc < 1)
    return LD_EINVAL;
  LD_LAUNCH(upsample_add_bwd_kernel, dim3((Hc * Wc + 255) / 256, rows),
                     dim3(256), 0, LD_STREAM, dout, Hf, Wf, Hc, Wc, addend, dcoarse);
  return (int)hipGetLastError();
}

extern "C" int ld_upsample_add_backward(const float* dout, int rows, int Hf, int Wf,
                                        int Hc, int Wc, float* dcoarse,
                                        ld_stream_t stream) {
  return ld_upsample_add_backward_acc(dout, rows, Hf, Wf, Hc, Wc, nullptr, dcoarse, stream);
}

namespace {
template <bool PACK>
int pack_levels_run(const ld_levels_t* lv, float* const* levels, int rows, float* x3,
                    ld_stream_t stream) {
  if (int e = check_levels(lv)) return e;
  if (!levels || !x3 || rows < 1) return LD_EINVAL;
  LevelPtrs lp{};
  for (int l = 0; l < lv->num_levels; ++l) {
    if (!levels[l]) return LD_EINVAL;
    lp.p[l] = levels[l];
  }
  const Levels k = make_levels(lv);
  LD_LAUNCH(pack_levels_kernel<PACK>, dim3((k.P + 255) / 256, rows), dim3(256), 0,
                     LD_STREAM, k, lp, x3);
  return (int)hipGetLastError();
}
}  // namespace

namespace {
template <bool PACK>
int pack_levels_c8_run(const ld_levels_t* lv, float* const* levels, int N, int C, float* x3,
                       void* const* levels_c8, void* x3_c8, ld_stream_t stream) {
  if (int e = check_levels(lv)) return e;
  if (!levels || !x3 || N < 1 || C < 8 || C % 8 != 0) return LD_EINVAL;
  if (PACK ? !x3_c8 : !levels_c8) return LD_EINVAL;
  LevelPtrs lp{};
  LevelC8Ptrs lc{};
  for (int l = 0; l < lv->num_levels; ++l) {
    if (!levels[l] || (!PACK && !levels_c8[l])) return LD_EINVAL;
    lp.p[l] = levels[l];
    if (!PACK) lc.p[l] = (pk_uintx4*)levels_c8[l];
  }
  const Levels k = make_levels(lv);
  LD_LAUNCH(pack_levels_c8_kernel<PACK>, dim3((k.P + 255) / 256, N * (C / 8)), dim3(256), 0,
            LD_STREAM, k, lp, C, x3, lc, (pk_uintx4*)x3_c8);
  return (int)hipGetLastError();
}
}  // namespace

extern "C" int ld_pack_levels_c8(const ld_levels_t* lv, const float* const* levels, int N,
                                 int C, float* x3, void* x3_c8, ld_stream_t stream) {
  return pack_levels_c8_run<true>(lv, const_cast<float* const*>(levels), N, C, x3, nullptr,
                                  x3_c8, stream);
}

extern "C" int ld_unpack_levels_c8(const ld_levels_t* lv, const float* x3, int N, int C,
                                   float* const* levels, void* const* levels_c8,
                                   ld_stream_t stream) {
  return pack_levels_c8_run<false>(lv, levels, N, C, const_cast<float*>(x3), levels_c8,
                                   nullptr, stream);
}

extern "C" int ld_pack_levels(const ld_levels_t* lv, const float* const* levels, int rows,
                              float* x3, ld_stream_t stream) {
  return pack_levels_run<true>(lv, const_cast<float* const*>(levels), rows, x3, stream);
}

extern "C" int ld_unpack_levels(const ld_levels_t* lv, const float* x3, int rows,
                                float* const* levels, ld_stream_t stream) {
  return pack_levels_run<false>(lv, levels, rows, const_cast<float*>(x3), stream);
}

extern "C" int ld_scale_levels_forward(const ld_levels_t* lv, const float* x,
                                       const float* scales, int rows, float* y,
                                       ld_stream_t stream) {
  if (int e = check_levels(lv)) return e;
  if (!x || !scales || !y || rows < 1) return LD_EINVAL;
  const Levels k = make_levels(lv);
  LD_LAUNCH(scale_levels_kernel, dim3((k.P + 255) / 256, rows), dim3(256),
                     0, LD_STREAM, x, k, scales, y);
  return (int)hipGetLastError();
}

extern "C" size_t ld_scale_levels_backward_workspace_bytes(const ld_levels_t* lv) {
  if (check_levels(lv) != 0) return 0;
  return (size_t)lv->num_levels * kScaleSplit * sizeof(double);
}

extern "C" int ld_scale_levels_backward(const ld_levels_t* lv, const float* dy,
                                        const float* x, const float* scales,
                                        int rows, float* dx, float* dscales,
                                        int accumulate, void* workspace,
                                        size_t workspace_bytes, ld_stream_t stream) {
  if (int e = check_levels(lv)) return e;
  if (!dy || !x || !scales || !dx || rows < 1) return LD_EINVAL;
  if (dscales && (!workspace ||
                  workspace_bytes < ld_scale_levels_backward_workspace_bytes(lv)))
    return LD_ENOSPACE;
  const Levels k = make_levels(lv);
  LD_LAUNCH(scale_levels_kernel, dim3((k.P + 255) / 256, rows), dim3(256),
                     0, LD_STREAM, dy, k, scales, dx);
  if (dscales) {
    LD_LAUNCH(scale_levels_bwd_partial_kernel,
                       dim3(k.num_levels * kScaleSplit), dim3(256), 0, LD_STREAM, dy,
                       x, k, rows, (double*)workspace);
    LD_LAUNCH(scale_levels_bwd_final_kernel, dim3(1), dim3(64), 0,
                       LD_STREAM, (const double*)workspace, k.num_levels, dscales,
                       accumulate);
  }
  return (int)hipGetLastError();
}

extern "C" int ld_sgd_step_dev(float* params, const float* grads, float* momentum_buf,
                               size_t n, const float* hyper, ld_stream_t stream) {
  if (!params || !grads || !momentum_buf || !hyper) return LD_EINVAL;
  if (n == 0) return 0;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum_buf) % 16)
    return LD_EINVAL;
  const size_t threads = (n + 3) / 4;
  LD_LAUNCH(sgd_dev_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256),
                     0, LD_STREAM, params, grads, momentum_buf, n, hyper);
  return (int)hipGetLastError();
}

extern "C" int ld_sgd_step(float* params, const float* grads, float* momentum_buf,
                           size_t n, float lr, float momentum, float weight_decay,
                           float grad_scale, ld_stream_t stream) {
  if (!params || !grads || !momentum_buf) return LD_EINVAL;
  if (n == 0) return 0;
  if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum_buf) % 16)
    return LD_EINVAL;
  const size_t threads = (n + 3) / 4;
  LD_LAUNCH(sgd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256),
                     0, LD_STREAM, params, grads, momentum_buf, n, lr, momentum,
                     weight_decay, grad_scale);
  return (int)hipGetLastError();
}
