// fp32 convolution for gfx950 as implicit GEMM on the f32-input matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 fmaf chains at the 157 TFLOP/s vector
// rate, MI355X_MICROARCH.md "Matrix cores").  Forward, data-gradient and
// weight-gradient of every conv on the LD train step: ResNet 1x1/3x3 (s1/s2),
// FPN lateral/output/extra convs, GFL head towers and predictors.
//
// Replaces nn.Conv2d (cuDNN/MIOpen) under
//   mmdet/models/backbones/resnet.py:35-46,163-183, utils/res_layer.py:38-59,
//   mmdet/models/necks/fpn.py:121-160, dense_heads/gfl_head.py:102-133.
//
// Data layout (chosen for MI355X, not inherited):
//   activations  NCHW fp32, optionally *level-concatenated*: (N, C, P) where
//                P = sum_l H_l*W_l holds all FPN levels of one image back to
//                back.  The GEMM column index j = n*P + p IS the output
//                address offset, so the epilogue stores 32 consecutive floats
//                per accumulator row (128-byte coalesced) and the five FPN
//                levels of a head conv (shared weights, gfl_head.py:175-180)
//                run as ONE launch instead of five under-filled ones.
//   weights      the module keeps mmdet's (Cout, Cin, KH, KW) parameter; a
//                transform kernel rewrites it per step into the GEMM-friendly
//                [tap][Cin][Cout] (forward) / [flipped tap][Cout][Cin] (dgrad)
//                images so A-tiles are contiguous 512-byte rows.
//   GEMM         D[co][j] = sum_{tap,ci} Wt[tap][ci][co] * X[ci][pos(j,tap)]
//                block tile BM(co) x 128(j) x 16(k), 4 wavefronts as 2x2, each
//                (BM/2)x64 = TM x 2 MFMA 32x32 tiles, double-buffered LDS with
//                the next tile's global loads issued under the MFMAs.
//   wgrad        D[co][ci] per tap = sum_j dY[co][j] * X[ci][pos(j,tap)],
//                split over j; partial slabs [split][tap][co][ci] are written
//                coalesced and summed in fixed order (deterministic, no float
//                atomics) by conv_wgrad_reduce.
#include <hip/hip_runtime.h>

#include "../../include/ld_hip.h"

namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

constexpr int kThreads = 256;
constexpr int BN = 128;   // GEMM columns (spatial) per block
constexpr int BK = 16;    // k-slice (input channels of one tap) per step
constexpr int WBK = 32;   // wgrad k-slice (spatial positions) per step
constexpr int WLD = WBK + 1;  // odd LDS row stride -> conflict-free columns

struct Geo {  // pyramid geometry as the gather sees it
  int stride, pad, num_levels;
  ld_conv_level_t lv[LD_MAX_LEVELS];
};

struct ConvK {  // kernel-side view of ld_conv_t + pointers
  const float* x;
  const float* wt;
  float* y;
  const float* bias;
  const float* scale;
  const float* shift;
  const float* residual;
  int relu;
  int N, Cin, Cout, KH, KW;
  int Pin, Pout;
  int J;  // N * Pout
  Geo g;
};

__device__ __forceinline__ int xcd_swizzle(int b, int nb) {
  // consecutive logical tiles -> same XCD (block b runs on XCD b % 8)
  const int q = nb >> 3, r = nb & 7;
  const int xcd = b & 7, idx = b >> 3;
  return xcd * q + min(xcd, r) + idx;
}

// position p in [0, Pout) -> level and (ho, wo)
__device__ __forceinline__ void locate_out(const Geo& a, int p, int& l, int& ho,
                                           int& wo) {
  l = 0;
#pragma unroll
  for (int i = 1; i < LD_MAX_LEVELS; ++i)
    if (i < a.num_levels && p >= a.lv[i].off_out) l = i;
  const int r = p - a.lv[l].off_out;
  ho = r / a.lv[l].Wout;
  wo = r - ho * a.lv[l].Wout;
}

// MODE 0: y = conv(x)            in = ho*S - P + kh
// MODE 1: transposed gather for the data-gradient of a stride-2 conv:
//         in position (ho - pad + kh) must be even; in = that / 2
template <int MODE>
__device__ __forceinline__ bool tap_offset(const Geo& a, int l, int ho, int wo,
                                           int kh, int kw, int& off) {
  const int Hin = a.lv[l].Hin, Win = a.lv[l].Win;
  int hi, wi;
  if (MODE == 0) {
    hi = ho * a.stride - a.pad + kh;
    wi = wo * a.stride - a.pad + kw;
  } else {
    const int hn = ho - a.pad + kh, wn = wo - a.pad + kw;
    if ((hn | wn) < 0 || ((hn | wn) & 1)) return false;
    hi = hn >> 1;
    wi = wn >> 1;
  }
  if (hi < 0 || hi >= Hin || wi < 0 || wi >= Win) return false;
  off = a.lv[l].off_in + hi * Win + wi;
  return true;
}

// ------------------------------------------------------------ forward/dgrad
template <int BM, int MODE>
__global__ __launch_bounds__(kThreads, 2) void conv_igemm_kernel(ConvK a) {
  constexpr int WM = BM / 2;       // wave tile rows
  constexpr int TM = WM / 32;      // MFMA tiles per wave along M
  constexpr int A_PER = BK * BM / kThreads;  // A floats per thread per step
  constexpr int B_PER = BK * BN / kThreads;  // = 8
  __shared__ float lds[2 * BK * (BM + BN)];
  float* As = lds;                  // [2][BK][BM]
  float* Bs = lds + 2 * BK * BM;    // [2][BK][BN]

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int mtiles = (a.Cout + BM - 1) / BM;
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (tile % mtiles) * BM;
  const int n0 = (tile / mtiles) * BN;

  // ---- this thread's B column (spatial position) -----------------------
  const int jb = n0 + (t & (BN - 1));
  const bool jvalid = jb < a.J;
  int bl = 0, bho = 0, bwo = 0;
  const float* xin = a.x;
  if (jvalid) {
    const int n = jb / a.Pout, p = jb - n * a.Pout;
    locate_out(a.g, p, bl, bho, bwo);
    xin = a.x + (size_t)n * a.Cin * a.Pin;
  }
  const int bk0 = (t / BN) * B_PER;  // first k row this thread loads for B
  // ---- this thread's A column (output channel) ------------------------
  const int am = t & (BM - 1);
  const bool avalid = (m0 + am) < a.Cout;
  const int ak0 = (t / BM) * A_PER;

  floatx16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ntaps = a.KH * a.KW;
  const int csteps = (a.Cin + BK - 1) / BK;
  const int ktot = a.Cin * ntaps;  // MODE 2: flat (ci, kh, kw) reduction index
  const int nsteps = (MODE == 2) ? (ktot + BK - 1) / BK : ntaps * csteps;
  float a_st[A_PER], b_st[B_PER];

  auto load_tile = [&](int step) {
    if (MODE == 2) {
      // small-Cin (stem) im2col: every k row has its own (ci, kh, kw)
      const int k0 = step * BK;
      const float* wp = a.wt + (size_t)(k0 + ak0) * a.Cout + m0 + am;
#pragma unroll
      for (int i = 0; i < A_PER; ++i)
        a_st[i] = (avalid && k0 + ak0 + i < ktot) ? wp[(size_t)i * a.Cout] : 0.0f;
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int k = k0 + bk0 + i;
        float v = 0.0f;
        if (jvalid && k < ktot) {
          const int ci = k / ntaps, r = k - ci * ntaps;
          const int kh = r / a.KW, kw = r - kh * a.KW;
          int off = 0;
          if (tap_offset<0>(a.g, bl, bho, bwo, kh, kw, off))
            v = xin[(size_t)ci * a.Pin + off];
        }
        b_st[i] = v;
      }
      return;
    }
    const int tap = step / csteps, ci0 = (step - tap * csteps) * BK;
    const int kh = tap / a.KW, kw = tap - kh * a.KW;
    int off = 0;
    const bool ok = jvalid && tap_offset<MODE>(a.g, bl, bho, bwo, kh, kw, off);
    const float* wp = a.wt + ((size_t)tap * a.Cin + ci0 + ak0) * a.Cout + m0 + am;
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      a_st[i] = (avalid && ci0 + ak0 + i < a.Cin) ? wp[(size_t)i * a.Cout] : 0.0f;
    const float* xp = xin + (size_t)(ci0 + bk0) * a.Pin + off;
#pragma unroll
    for (int i = 0; i < B_PER; ++i)
      b_st[i] = (ok && ci0 + bk0 + i < a.Cin) ? xp[(size_t)i * a.Pin] : 0.0f;
  };
  auto store_tile = [&](int buf) {
    float* ap = As + buf * BK * BM + ak0 * BM + am;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) ap[i * BM] = a_st[i];
    float* bp = Bs + buf * BK * BN + bk0 * BN + (t & (BN - 1));
#pragma unroll
    for (int i = 0; i < B_PER; ++i) bp[i * BN] = b_st[i];
  };

  load_tile(0);
  store_tile(0);
  __syncthreads();
  const int l31 = lane & 31, lk = lane >> 5;
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    if (step + 1 < nsteps) load_tile(step + 1);
    const float* ap = As + cur * BK * BM + wm * WM + l31;
    const float* bp = Bs + cur * BK * BN + wn * 64 + l31;
#pragma unroll
    for (int kp = 0; kp < BK / 2; ++kp) {
      const int kr = 2 * kp + lk;
      float af[TM], bf[2];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = ap[kr * BM + i * 32];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = bp[kr * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j],
                                                           0, 0, 0);
    }
    if (step + 1 < nsteps) store_tile(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int jc = n0 + wn * 64 + j * 32 + l31;
    if (jc >= a.J) continue;
    const int n = jc / a.Pout, p = jc - n * a.Pout;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (co >= a.Cout) continue;
        const size_t idx = ((size_t)n * a.Cout + co) * a.Pout + p;
        float v = acc[i][j][r];
        if (a.scale) v = v * a.scale[co] + a.shift[co];
        if (a.bias) v += a.bias[co];
        if (a.residual) v += a.residual[idx];
        if (a.relu) v = fmaxf(v, 0.0f);
        a.y[idx] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ wgrad --
struct WgradK {
  const float* x;    // (N, Cin, Pin)
  const float* dy;   // (N, Cout, Pout)
  float* slabs;      // [split][tap][Cout][Cin]
  int N, Cin, Cout, KH, KW;
  int Pin, Pout;
  int J, splits, jchunk;  // jchunk: columns per split (multiple of WBK)
  Geo g;
};

__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_kernel(WgradK a) {
  constexpr int BM = 128, BNc = 128;  // co x ci tile
  constexpr int ROWS_PER = BM / 8;    // rows per thread (8 row-groups of 32 lanes)
  __shared__ float lds[2 * (BM + BNc) * WLD];
  float* As = lds;                       // [2][BM][WLD]   dY rows (co), k = j
  float* Bs = lds + 2 * BM * WLD;        // [2][BNc][WLD]  X rows (ci), k = j

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int mt = (a.Cout + BM - 1) / BM, nt = (a.Cin + BNc - 1) / BNc;
  const int ntaps = a.KH * a.KW;
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int ntile = b % nt;
  b /= nt;
  const int mtile = b % mt;
  b /= mt;
  const int tap = b % ntaps;
  const int split = b / ntaps;
  const int m0 = mtile * BM, c0 = ntile * BNc;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int jbeg = split * a.jchunk;
  const int jend = min(a.J, jbeg + a.jchunk);

  const int kq = t & (WBK - 1);  // this thread's k (column j offset) in a step
  const int r0 = t >> 5;         // 0..7: first row; rows r0 + 8*i

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  float a_st[ROWS_PER], b_st[ROWS_PER];
  auto load_tile = [&](int j0) {
    const int j = j0 + kq;
    bool jok = j < jend;
    int off = 0;
    size_t ybase = 0, xbase = 0;
    bool xok = false;
    if (jok) {
      const int n = j / a.Pout, p = j - n * a.Pout;
      int l, ho, wo;
      locate_out(a.g, p, l, ho, wo);
      ybase = (size_t)n * a.Cout * a.Pout + p;
      xbase = (size_t)n * a.Cin * a.Pin;
      xok = tap_offset<0>(a.g, l, ho, wo, kh, kw, off);
    }
#pragma unroll
    for (int i = 0; i < ROWS_PER; ++i) {
      const int co = m0 + r0 + 8 * i;
      a_st[i] = (jok && co < a.Cout) ? a.dy[ybase + (size_t)co * a.Pout] : 0.0f;
      const int ci = c0 + r0 + 8 * i;
      b_st[i] = (xok && ci < a.Cin) ? a.x[xbase + (size_t)ci * a.Pin + off] : 0.0f;
    }
  };
  auto store_tile = [&](int buf) {
    float* ap = As + buf * BM * WLD + r0 * WLD + kq;
    float* bp = Bs + buf * BNc * WLD + r0 * WLD + kq;
#pragma unroll
    for (int i = 0; i < ROWS_PER; ++i) {
      ap[8 * i * WLD] = a_st[i];
      bp[8 * i * WLD] = b_st[i];
    }
  };

  const int nsteps = (jend - jbeg + WBK - 1) / WBK;
  const int l31 = lane & 31, lk = lane >> 5;
  if (nsteps > 0) {
    load_tile(jbeg);
    store_tile(0);
  }
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    if (step + 1 < nsteps) load_tile(jbeg + (step + 1) * WBK);
    const float* ap = As + cur * BM * WLD + (wm * 64 + l31) * WLD;
    const float* bp = Bs + cur * BNc * WLD + (wn * 64 + l31) * WLD;
#pragma unroll
    for (int kp = 0; kp < WBK / 2; ++kp) {
      const int kc = 2 * kp + lk;
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = ap[i * 32 * WLD + kc];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = bp[j * 32 * WLD + kc];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j],
                                                           0, 0, 0);
    }
    if (step + 1 < nsteps) store_tile(cur ^ 1);
    __syncthreads();
  }
  // slab store: [split][tap][co][ci], ci fastest (= lane & 31)
  float* slab = a.slabs + ((size_t)split * ntaps + tap) * a.Cout * a.Cin;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ci = c0 + wn * 64 + j * 32 + l31;
    if (ci >= a.Cin) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (co < a.Cout) slab[(size_t)co * a.Cin + ci] = acc[i][j][r];
      }
  }
}

// dW[co][ci][tap] (+)= sum_split slab[split][tap][co][ci]
__global__ void conv_wgrad_reduce_kernel(const float* __restrict__ slabs, int splits,
                                         int ntaps, int Cout, int Cin,
                                         float* __restrict__ dw, int accumulate) {
  const size_t per = (size_t)ntaps * Cout * Cin;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per) return;
  // i enumerates [tap][co][ci] (coalesced reads)
  const int ci = (int)(i % Cin);
  const size_t q = i / Cin;
  const int co = (int)(q % Cout), tap = (int)(q / Cout);
  float s = 0.0f;
  for (int k = 0; k < splits; ++k) s += slabs[(size_t)k * per + i];
  const size_t o = ((size_t)co * Cin + ci) * ntaps + tap;
  dw[o] = accumulate ? dw[o] + s : s;
}

// (Cout, Cin, KH, KW) -> fwd image [tap][Cin][Cout] and dgrad image
// [KH*KW-1-tap][Cout][Cin]
__global__ void conv_weight_transform_kernel(const float* __restrict__ w, int Cout,
                                             int Cin, int ntaps,
                                             float* __restrict__ wt_fwd,
                                             float* __restrict__ wt_bwd) {
  const size_t total = (size_t)Cout * Cin * ntaps;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  if (wt_fwd) {  // i enumerates the fwd image [tap][ci][co] (coalesced writes)
    const int co = (int)(i % Cout);
    const size_t q = i / Cout;
    const int ci = (int)(q % Cin), tap = (int)(q / Cin);
    wt_fwd[i] = w[((size_t)co * Cin + ci) * ntaps + tap];
  }
  if (wt_bwd) {  // i enumerates the bwd image [tapf][co][ci]
    const int ci = (int)(i % Cin);
    const size_t q = i / Cin;
    const int co = (int)(q % Cout), tapf = (int)(q / Cout);
    wt_bwd[i] = w[((size_t)co * Cin + ci) * ntaps + (ntaps - 1 - tapf)];
  }
}

int check_conv(const ld_conv_t* c) {
  if (!c || c->N < 1 || c->Cin < 1 || c->Cout < 1 || c->KH < 1 || c->KW < 1 ||
      c->num_levels < 1 || c->num_levels > LD_MAX_LEVELS)
    return LD_EINVAL;
  if (c->stride != 1 && c->stride != 2) return LD_EUNSUPPORTED;
  int pin = 0, pout = 0;
  for (int l = 0; l < c->num_levels; ++l) {
    const ld_conv_level_t& v = c->lv[l];
    if (v.off_in != pin || v.off_out != pout) return LD_EINVAL;
    if ((v.Hin + 2 * c->pad - c->KH) / c->stride + 1 != v.Hout ||
        (v.Win + 2 * c->pad - c->KW) / c->stride + 1 != v.Wout)
      return LD_EINVAL;
    pin += v.Hin * v.Win;
    pout += v.Hout * v.Wout;
  }
  if (pin != c->Pin || pout != c->Pout) return LD_EINVAL;
  return 0;
}

template <int MODE>
int launch_igemm(const ConvK& k, hipStream_t stream) {
  const int J = k.J;
  const int ntile = (J + BN - 1) / BN;
  if (k.Cout <= 64) {
    const int mt = (k.Cout + 63) / 64;
    hipLaunchKernelGGL((conv_igemm_kernel<64, MODE>), dim3(mt * ntile),
                       dim3(kThreads), 0, stream, k);
  } else {
    const int mt = (k.Cout + 127) / 128;
    hipLaunchKernelGGL((conv_igemm_kernel<128, MODE>), dim3(mt * ntile),
                       dim3(kThreads), 0, stream, k);
  }
  return (int)hipGetLastError();
}

}  // namespace

extern "C" int ld_conv_weight_transform(const float* w, int Cout, int Cin, int KH,
                                        int KW, float* wt_fwd, float* wt_bwd,
                                        ld_stream_t stream) {
  if (!w || Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || (!wt_fwd && !wt_bwd))
    return LD_EINVAL;
  const size_t total = (size_t)Cout * Cin * KH * KW;
  hipLaunchKernelGGL(conv_weight_transform_kernel,
                     dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, Cout, Cin, KH * KW, wt_fwd, wt_bwd);
  return (int)hipGetLastError();
}

extern "C" int ld_conv_forward(const ld_conv_t* c, const float* x,
                               const float* wt_fwd, const ld_conv_epilogue_t* ep,
                               float* y, ld_stream_t stream) {
  if (int e = check_conv(c)) return e;
  if (!x || !wt_fwd || !y) return LD_EINVAL;
  ConvK k;
  k.x = x;
  k.wt = wt_fwd;
  k.y = y;
  k.bias = ep ? ep->bias : nullptr;
  k.scale = ep ? ep->scale : nullptr;
  k.shift = ep ? ep->shift : nullptr;
  k.residual = ep ? ep->residual : nullptr;
  k.relu = ep ? ep->relu : 0;
  if ((k.scale == nullptr) != (k.shift == nullptr)) return LD_EINVAL;
  k.N = c->N; k.Cin = c->Cin; k.Cout = c->Cout; k.KH = c->KH; k.KW = c->KW;
  k.g.stride = c->stride; k.g.pad = c->pad; k.Pin = c->Pin; k.Pout = c->Pout;
  k.g.num_levels = c->num_levels;
  k.J = c->N * c->Pout;
  for (int l = 0; l < LD_MAX_LEVELS; ++l) k.g.lv[l] = c->lv[l];
  return launch_igemm<0>(k, (hipStream_t)stream);
}

// Small-Cin convolution (the 7x7 stem, Cin = 3): the reduction index is the
// flat (ci, kh, kw) triple; wt is the [Cin*KH*KW][Cout] image, i.e.
// ld_conv_weight_transform(w, Cout, Cin*KH*KW, 1, 1, wt, NULL).
extern "C" int ld_conv_forward_smallc(const ld_conv_t* c, const float* x,
                                      const float* wt, const ld_conv_epilogue_t* ep,
                                      float* y, ld_stream_t stream) {
  if (int e = check_conv(c)) return e;
  if (!x || !wt || !y) return LD_EINVAL;
  ConvK k;
  k.x = x;
  k.wt = wt;
  k.y = y;
  k.bias = ep ? ep->bias : nullptr;
  k.scale = ep ? ep->scale : nullptr;
  k.shift = ep ? ep->shift : nullptr;
  k.residual = ep ? ep->residual : nullptr;
  k.relu = ep ? ep->relu : 0;
  if ((k.scale == nullptr) != (k.shift == nullptr)) return LD_EINVAL;
  k.N = c->N; k.Cin = c->Cin; k.Cout = c->Cout; k.KH = c->KH; k.KW = c->KW;
  k.g.stride = c->stride; k.g.pad = c->pad; k.Pin = c->Pin; k.Pout = c->Pout;
  k.g.num_levels = c->num_levels;
  k.J = c->N * c->Pout;
  for (int l = 0; l < LD_MAX_LEVELS; ++l) k.g.lv[l] = c->lv[l];
  return launch_igemm<2>(k, (hipStream_t)stream);
}

// dx = conv_transpose(dy): runs the same implicit GEMM with the roles of the
// channel dims swapped, the flipped-tap weight image and pad' = K - 1 - pad.
extern "C" int ld_conv_dgrad(const ld_conv_t* c, const float* dy,
                             const float* wt_bwd, float* dx, ld_stream_t stream) {
  if (int e = check_conv(c)) return e;
  if (!dy || !wt_bwd || !dx) return LD_EINVAL;
  if (c->KH != c->KW) return LD_EUNSUPPORTED;
  ConvK k;
  k.x = dy;
  k.wt = wt_bwd;
  k.y = dx;
  k.bias = k.scale = k.shift = k.residual = nullptr;
  k.relu = 0;
  k.N = c->N; k.Cin = c->Cout; k.Cout = c->Cin; k.KH = c->KH; k.KW = c->KW;
  k.g.stride = 1;
  k.g.pad = c->KH - 1 - c->pad;
  k.Pin = c->Pout; k.Pout = c->Pin;
  k.g.num_levels = c->num_levels;
  k.J = c->N * c->Pin;
  for (int l = 0; l < LD_MAX_LEVELS; ++l) {
    k.g.lv[l].Hin = c->lv[l].Hout; k.g.lv[l].Win = c->lv[l].Wout;
    k.g.lv[l].Hout = c->lv[l].Hin; k.g.lv[l].Wout = c->lv[l].Win;
    k.g.lv[l].off_in = c->lv[l].off_out; k.g.lv[l].off_out = c->lv[l].off_in;
  }
  if (c->stride == 1) return launch_igemm<0>(k, (hipStream_t)stream);
  return launch_igemm<1>(k, (hipStream_t)stream);
}

static int wgrad_splits(const ld_conv_t* c) {
  const int J = c->N * c->Pout;
  const int tiles = ((c->Cout + 127) / 128) * ((c->Cin + 127) / 128) * c->KH * c->KW;
  int splits = (1024 + tiles - 1) / tiles;            // aim at ~1024 blocks
  const int max_by_k = (J + 8 * WBK - 1) / (8 * WBK); // >= 8 steps per block
  if (splits > max_by_k) splits = max_by_k;
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  return splits;
}

extern "C" size_t ld_conv_wgrad_workspace_bytes(const ld_conv_t* c) {
  if (check_conv(c) != 0) return 0;
  return (size_t)wgrad_splits(c) * c->KH * c->KW * c->Cout * c->Cin * sizeof(float);
}

extern "C" int ld_conv_wgrad(const ld_conv_t* c, const float* x, const float* dy,
                             float* dw, int accumulate, void* workspace,
                             size_t workspace_bytes, ld_stream_t stream_) {
  if (int e = check_conv(c)) return e;
  if (!x || !dy || !dw) return LD_EINVAL;
  if (!workspace || workspace_bytes < ld_conv_wgrad_workspace_bytes(c))
    return LD_ENOSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  WgradK k;
  k.x = x; k.dy = dy; k.slabs = (float*)workspace;
  k.N = c->N; k.Cin = c->Cin; k.Cout = c->Cout; k.KH = c->KH; k.KW = c->KW;
  k.g.stride = c->stride; k.g.pad = c->pad; k.Pin = c->Pin; k.Pout = c->Pout;
  k.g.num_levels = c->num_levels;
  k.J = c->N * c->Pout;
  k.splits = wgrad_splits(c);
  int jchunk = (k.J + k.splits - 1) / k.splits;
  jchunk = (jchunk + WBK - 1) / WBK * WBK;
  k.jchunk = jchunk;
  for (int l = 0; l < LD_MAX_LEVELS; ++l) k.g.lv[l] = c->lv[l];
  const int ntaps = c->KH * c->KW;
  const int blocks = ((c->Cout + 127) / 128) * ((c->Cin + 127) / 128) * ntaps * k.splits;
  hipLaunchKernelGGL(conv_wgrad_kernel, dim3(blocks), dim3(kThreads), 0, stream, k);
  const size_t per = (size_t)ntaps * c->Cout * c->Cin;
  hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)((per + 255) / 256)),
                     dim3(256), 0, stream, k.slabs, k.splits, ntaps, c->Cout, c->Cin,
                     dw, accumulate);
  return (int)hipGetLastError();
}
