// bf16 convolution for gfx950: the same implicit GEMMs as conv.hip on
// v_mfma_f32_32x32x16_bf16 (16x the f32-MFMA rate, fp32 accumulate) -- the
// compute path of BASELINE.json config 3 (ld_r50_gflv1_r101_fpn_coco_1x, bf16).
// Reference: the fp16 mode of the same layers (mmcv auto_fp16 around
// backbone / neck / head forward; the head's reg output is cast back with
// .float() at mmdet/models/dense_heads/gfl_head.py:181-183 and the loss block
// is @force_fp32, ld_head.py:284): low-precision matrix operands, fp32
// everything else.
//
// Precision contract of this file: operands are rounded to bf16 (RNE) on the
// way into the matrix core, products are exact, accumulation is fp32 in one
// accumulator per output element; master weights, activations in HBM,
// epilogue arithmetic (BN affine, bias, residual, ReLU) and all gradients
// stay fp32.
//
// Layouts
//   activations  (N, C, P) fp32, as conv.hip (level-concatenated where needed)
//   weight image bf16 [tap][K/8][Cout][8]: the eight k (input channels) an MFMA
//                lane feeds are ONE 16-byte load; K = Cin rounded up to 16, zero
//                filled.  The dgrad image swaps the channel roles and flips taps.
//   fwd/dgrad    conv_stream_bf16_kernel: LDS-free like conv_stream_kernel.  The
//                32x32x16 operand layouts are  A lane l <- Wt[tap][ci/8 + (l>>5)][co0 + (l&31)][0..7]
//                (16 B per lane, 512 B contiguous per half-wave) and
//                B lane l <- X[n][ci + 8*(l>>5) + e][pos(j0 + (l&31))], e = 0..7:
//                eight 128-byte rows per half-wave, converted in registers
//                (v_cvt_pk_bf16_f32).  A D-deep register ring of k16-steps per
//                wavefront, no barriers (KS = 4: one LDS meeting before the
//                epilogue), shapes picked from the shared tuning table.
//   wgrad        conv_wgrad_wave_bf16_kernel: wave-private 64x64 tile; operand
//                tiles go global (coalesced along j) -> bf16 -> PRIVATE LDS
//                [row][j] with an 80-byte row pitch -> 16-byte fragment reads
//                (conflict-free), 8 MFMAs per 32 positions.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float floatx8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));

// Epilogue side output (a.y_c8): the bf16 channel-blocked (C8) image of y, for
// the conv that consumes y next.  A lane holds rows rbase + 8g + {0..3} (g =
// 0..3) of one position: four consecutive channels = one half of a 16-byte C8
// row -> one 8-byte store; the two lane halves of a wave fill the row,
// consecutive lanes consecutive positions (fully coalesced).

inline int k8_blocks(int k) { return (k + 15) / 16 * 2; }  // 8-blocks per tap

__device__ __forceinline__ uintx4 buf_load16(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(uintx4,
                            __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

// ------------------------------------------------- forward/dgrad, streaming
template <int TM, int TN, int WVM, int MODE, int D, int OCC, int KS>
__global__ __launch_bounds__(256, OCC) void conv_stream_bf16_kernel(ConvK a) {
  static_assert(KS == 1 || (KS == 4 && WVM == 1), "KS is 1 or 4");
  static_assert(D * (TM + 8 * TN) < 64, "ring exceeds the vmcnt range");
  constexpr int WVN = 4 / WVM;
  constexpr int WM = TM * 32, WN = TN * 32;
  constexpr int BM = KS == 4 ? WM : WVM * WM, BNT = KS == 4 ? WN : WVN * WN;
  __shared__ float red[KS == 4 ? 2 * TM * TN * 16 * 64 : 1];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = KS == 4 ? 0 : wave / WVN, wn = KS == 4 ? 0 : wave % WVN;
  const int kslice = KS == 4 ? wave : 0;
  const int l31 = lane & 31, lk = lane >> 5;
  const int mtiles = (a.Cout + BM - 1) / BM;
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (tile % mtiles) * BM + wm * WM;
  const int n0 = (tile / mtiles) * BNT + wn * WN;
  // KS == 1: no barriers below, waves are independent.  KS == 4: the condition
  // is uniform over the workgroup.
  if (m0 >= a.Cout || n0 >= a.J) return;

  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int KW = __builtin_amdgcn_readfirstlane(a.KW);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Kp8 = __builtin_amdgcn_readfirstlane(a.Kpad);  // 8-blocks per tap
  const int ntw = __builtin_amdgcn_readfirstlane(MODE == 1 ? a.ntw : a.KW);
  const int ntaps = __builtin_amdgcn_readfirstlane(
      MODE == 1 ? a.nth * a.ntw : a.KH * a.KW);

  // ---- per-lane column geometry, resolved once ----------------------------
  int bHin[TN], bWin[TN], boff[TN], bh0[TN], bw0[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int jb = n0 + j * 32 + l31;
    bHin[j] = 0;  // Hin = 0 -> every tap out of range -> kOOB -> zeros
    bWin[j] = 0;
    boff[j] = 0;
    bh0[j] = 0;
    bw0[j] = 0;
    if (jb < a.J) {
      const int n = jb / a.Pout, p = jb - n * a.Pout;
      int bl, bho, bwo;
      locate_out(a.g, p, bl, bho, bwo);
      bHin[j] = a.g.lv[bl].Hin;
      bWin[j] = a.g.lv[bl].Win;
      // this lane's eight k rows start at channel 8 * lk of the k16-step
      boff[j] = n * Cin * Pin + a.g.lv[bl].off_in + 8 * lk * Pin;
      if (MODE == 1) {
        bh0[j] = bho + a.ch0;
        bw0[j] = bwo + a.cw0;
      } else {
        bh0[j] = bho * a.g.stride - a.g.pad;
        bw0[j] = bwo * a.g.stride - a.g.pad;
      }
    }
  }
  unsigned va[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int co = m0 + i * 32 + l31;
    va[i] = co < Cout ? (unsigned)(lk * Cout + co) * 16u : kOOB;
  }
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t rw = make_rsrc(a.wt, a.wt_bytes);

  unsigned vb[TN];
  int wtap = 0;  // weight-image tap of the cursor
  auto set_tap = [&](int tap) {
    int kh = tap / ntw, kw = tap - kh * ntw;
    if (MODE == 1) {
      wtap = (a.kh0 + 2 * kh) * KW + a.kw0 + 2 * kw;
    } else {
      wtap = tap;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int hi = bh0[j] + kh, wi = bw0[j] + kw;
      const bool ok = hi >= 0 && hi < bHin[j] && wi >= 0 && wi < bWin[j];
      vb[j] = ok ? (unsigned)(boff[j] + hi * bWin[j] + wi) * 4u : kOOB;
    }
  };
  uintx4 ra[D][TM];
  float rb[D][TN][8];
  // one k16-step: sa = byte offset of 8-block (tap, ci/8) in the weight image,
  // sb = byte offset of channel row ci in the activation tensor
  auto load_k16 = [&](int d, unsigned sa, unsigned sb) {
#pragma unroll
    for (int i = 0; i < TM; ++i) ra[d][i] = buf_load16(rw, va[i], sa);
    const unsigned prow = (unsigned)Pin * 4u;
    unsigned so = sb;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
#pragma unroll
      for (int j = 0; j < TN; ++j) rb[d][j][e] = buf_load(rx, vb[j], so);
      so += prow;
    }
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  auto mfma_k16 = [&](int d) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      floatx8 f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = rb[d][j][e];
      const bf16x8 b = __builtin_convertvector(f, bf16x8);
#pragma unroll
      for (int i = 0; i < TM; ++i)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            __builtin_bit_cast(bf16x8, ra[d][i]), b, acc[i][j], 0, 0, 0);
    }
  };

  const int cchunks = (Cin >> 4) / D;  // host guarantees (Cin / 16) % D == 0
  const int nchunks = ntaps * cchunks;
  // this wave's chunks: kslice, kslice + KS, ...
  const int mychunks = nchunks > kslice ? (nchunks - kslice + KS - 1) / KS : 0;
  int ltap = 0, lci = 0;
  auto advance = [&](int n) {
    lci += 16 * D * n;
    const int t0 = ltap;
    while (lci >= Cin) {
      lci -= Cin;
      ++ltap;
    }
    if (ltap != t0 && ltap < ntaps) set_tap(ltap);
  };
  set_tap(0);
  if (KS > 1) advance(kslice);
  const unsigned astep = (unsigned)(2 * Cout) * 16u;  // two 8-blocks per k16-step
  const unsigned bstep = (unsigned)(16 * Pin) * 4u;
  if (mychunks > 0) {
    {
      const unsigned sa = (unsigned)((wtap * Kp8 + (lci >> 3)) * Cout) * 16u;
      const unsigned sb = (unsigned)lci * Pin * 4u;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        load_k16(d, sa + (unsigned)d * astep, sb + (unsigned)d * bstep);
        __builtin_amdgcn_sched_barrier(0);
      }
      advance(KS);
    }
    for (int c = 0; c + 1 < mychunks; ++c) {
      const unsigned sa = (unsigned)((wtap * Kp8 + (lci >> 3)) * Cout) * 16u;
      const unsigned sb = (unsigned)lci * Pin * 4u;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        // pin the ring order (see conv_stream_kernel): MFMAs of slot d, then
        // the refill of slot d, so D - 1 slots of loads stay in flight
        mfma_k16(d);
        __builtin_amdgcn_sched_barrier(0);
        load_k16(d, sa + (unsigned)d * astep, sb + (unsigned)d * bstep);
        __builtin_amdgcn_sched_barrier(0);
      }
      advance(KS);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) mfma_k16(d);
  }
  if (KS == 4) {
    // pairwise tree through LDS, fixed order ((w0 + w2) + (w1 + w3))
    constexpr int NACC = TM * TN * 16;
    auto put = [&](float* dst) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dst[((i * TN + j) * 16 + r) * 64 + lane] = acc[i][j][r];
    };
    auto add = [&](const float* src) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            acc[i][j][r] += src[((i * TN + j) * 16 + r) * 64 + lane];
    };
    if (wave >= 2) put(red + (size_t)(wave - 2) * NACC * 64);
    __syncthreads();
    if (wave < 2) add(red + (size_t)wave * NACC * 64);
    __syncthreads();
    if (wave == 1) put(red);
    __syncthreads();
    if (wave != 0) return;
    add(red);
  }

  // ---- epilogue: direct stores, 32 consecutive positions per accumulator row
  const bool has_res = a.residual != nullptr;
  const bool res8 = a.res_c8 != nullptr;  // residual as a C8 image
  const bool has_y = a.y != nullptr;      // fp32 output optional (C8-only nets)
  const bool c8out = a.y_c8 != nullptr;
  const bool relu = a.relu != 0;
  const bool has_aff = a.scale != nullptr, has_bias = a.bias != nullptr;
  // Loop order (round 3): row group g = 4 accumulator rows (one C8 half-row) x
  // all TN column tiles; every load of a group -- affine, bias, fp32 / C8
  // residual -- is issued before the group's first store.  The pointers of ConvK
  // may alias as far as hipcc knows, so a residual load written between stores
  // stays behind the previous store: the loop this replaces made up to 16
  // dependent round trips per 32 x 32 tile, each waited for with vmcnt(0) (ISA
  // reading; the 256 -> 1024 expansion convs ran at 1.2 TB/s).
  size_t colbase[TN], c8base[TN];
  bool jok[TN];
  const int prow = MODE == 1 ? a.Pfull : a.Pout;
  // per-channel operands through descriptors: an absent one has extent 0 and
  // loads as 0 -- no flag-dependent branch inside a group's load phase
  const rsrc_t r_sc = make_rsrc(a.scale, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_sh = make_rsrc(a.shift, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_bi = make_rsrc(a.bias, has_bias ? (unsigned)Cout * 4u : 0u);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int jc = n0 + j * 32 + l31;
    jok[j] = jc < a.J;
    const int n = jc / a.Pout;
    int p = jc - n * a.Pout;
    // C8 images (side output, residual) exist for MODE 0 only: compact p
    c8base[j] = ((size_t)n * (Cout >> 3) * a.Pout + p) * 16;
    if (MODE == 1 && jok[j]) {
      int l, hc, wc;
      locate_out(a.g, p, l, hc, wc);
      p = a.foff[l] + (2 * hc + a.ph) * a.fW[l] + 2 * wc + a.pw;
    }
    colbase[j] = (size_t)n * Cout * prow + p;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rbase = m0 + i * 32 + 4 * lk;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row0 = rbase + 8 * g;  // this lane's four rows: row0 .. row0 + 3
      // byte offset of the lane's 8-byte half of its C8 row within an image
      const size_t c8row = (size_t)(row0 >> 3) * a.Pout * 16 + (row0 & 4) * 2;
      float sc[4], sh[4], bi[4], rv[TN][4];
      uintx2 rraw[TN];
      // group fence: the loads of this group stay behind the previous group's
      // stores and none of them is consumed before all are issued
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned ro = (unsigned)(row0 + e) * 4u;  // >= Cout / absent: zeros
        sc[e] = buf_load(r_sc, ro, 0);
        sh[e] = buf_load(r_sh, ro, 0);
        bi[e] = buf_load(r_bi, ro, 0);
      }
      if (has_res) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            rv[j][e] = (jok[j] && row0 + e < Cout)
                           ? a.residual[colbase[j] + (size_t)(row0 + e) * prow]
                           : 0.0f;
      }
      if (MODE == 0 && res8) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          rraw[j] = (jok[j] && row0 < Cout)
                        ? *(const uintx2*)((const char*)a.res_c8 + c8base[j] + c8row)
                        : uintx2{0u, 0u};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!jok[j]) continue;
        floatx4_t q;
        floatx4_t rq = {0.0f, 0.0f, 0.0f, 0.0f};
        if (MODE == 0 && res8)
          rq = __builtin_convertvector(__builtin_bit_cast(bf16x4, rraw[j]), floatx4_t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int row = row0 + e;
          float v = 0.0f;
          if (row < Cout) {
            v = acc[i][j][r] * (has_aff ? sc[e] : 1.0f) + (sh[e] + bi[e]);
            if (MODE == 0 && a.y_raw)
              a.y_raw[colbase[j] + (size_t)row * prow] = acc[i][j][r];
            if (has_res) v += rv[j][e];
            if (MODE == 0 && res8) v += rq[e];
            if (relu) v = fmaxf(v, 0.0f);
            if (has_y) a.y[colbase[j] + (size_t)row * prow] = v;
          }
          q[e] = v;
        }
        // side output: this lane's four channels = half of a 16-byte C8 row
        if (MODE == 0 && c8out && row0 < Cout)
          *(uintx2*)((char*)a.y_c8 + c8base[j] + c8row) =
              __builtin_bit_cast(uintx2, __builtin_convertvector(q, bf16x4));
      }
    }
  }
}

// ---------------------------------------------- forward/dgrad, LDS-tiled --
// The streaming kernel above loads every operand element once PER WAVEFRONT and
// is capped by the CU's vector-cache rate at bf16 MFMA speed (16x the f32 rate:
// a 2x2 wave tile asks for ~190 B/clk/CU, the L1 delivers ~64).  Here the four
// wavefronts of a workgroup share one (BM x BN) tile: every element is fetched
// from L1/L2 ONCE per workgroup, rounded to bf16 on the way into LDS, and read
// back as MFMA fragments by the two waves that need it.
//   LDS image (both operands) [BK/8][rows][8] bf16: the eight k of one row are 16
//   contiguous bytes -- the A image is a straight copy of the weight image rows,
//   the B image is built from 16 (BN = 128) coalesced fp32 row loads per thread,
//   packed with v_cvt_pk_bf16_f32 and stored as ds_write_b128; a fragment is ONE
//   ds_read_b128 whose 32 lanes x 16 B are consecutive (conflict-free, no swizzle
//   needed because the row pitch is 16 B).
//   Pipeline: at bf16 MFMA speed one 32-deep k-step is only 256 matrix cycles
//   per wave while a load takes ~1-2 k cycles to come back, and most layers of
//   the step yield only ~1 workgroup per CU -- so the loads run NST - 1 steps
//   AHEAD through a ring of NST register stages (statically indexed: the step
//   loop is unrolled by NST), two LDS buffers, one barrier per step:
//     issue loads(step + NST - 1) -> convert + ds_write stage(step) -> barrier
//     -> fragments + MFMAs of step.
//   (measured: with the loads only one step ahead the 128 x 128 tile reached
//   261 TFLOP/s on the 100x168 stage, 12 % of the pipe, profiles/
//   r02_kernels_s3_bf16.json.)
//   Waves 2 x 2, wave tile (BM/2) x (BN/2) = TM x TN MFMA 32x32x16 tiles.
template <int BM, int BN, int MODE, int NST>
__global__ __launch_bounds__(256, 2) void conv_tile_bf16_kernel(ConvK a) {
  static_assert(NST % 2 == 0, "ring length must be even (static LDS parity)");
  constexpr int BK = 32, KB = BK / 8;      // k rows per step, 8-blocks per step
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_U = BM * KB / 256;       // 16-byte units of A per thread
  constexpr int CG = 256 / BN;             // channel groups across the block
  constexpr int CPT = BK / CG;             // channels per thread (8 or 16)
  static_assert(CPT % 8 == 0 && A_U >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) uintx4 lds[2 * KB * (BM + BN)];
  uintx4* As = lds;                   // [2][KB][BM]
  uintx4* Bs = lds + 2 * KB * BM;     // [2][KB][BN]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lk = lane >> 5;
  const int mtiles = (a.Cout + BM - 1) / BM;
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (tile % mtiles) * BM;
  const int n0 = (tile / mtiles) * BN;

  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int KW = __builtin_amdgcn_readfirstlane(a.KW);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Kp8 = __builtin_amdgcn_readfirstlane(a.Kpad);
  const int ntw = __builtin_amdgcn_readfirstlane(MODE == 1 ? a.ntw : a.KW);
  const int ntaps = __builtin_amdgcn_readfirstlane(
      MODE == 1 ? a.nth * a.ntw : a.KH * a.KW);

  // ---- this thread's B column (one spatial position) ----------------------
  const int bp = t % BN, cg = t / BN;
  int bHin = 0, bWin = 0, boff = 0, bh0 = 0, bw0 = 0;
  {
    const int jb = n0 + bp;
    if (jb < a.J) {
      const int n = jb / a.Pout, p = jb - n * a.Pout;
      int bl, bho, bwo;
      locate_out(a.g, p, bl, bho, bwo);
      bHin = a.g.lv[bl].Hin;
      bWin = a.g.lv[bl].Win;
      boff = n * Cin * Pin + a.g.lv[bl].off_in + cg * CPT * Pin;
      if (MODE == 1) {
        bh0 = bho + a.ch0;
        bw0 = bwo + a.cw0;
      } else {
        bh0 = bho * a.g.stride - a.g.pad;
        bw0 = bwo * a.g.stride - a.g.pad;
      }
    }
  }
  // ---- this thread's A units: unit u -> (kb = u / BM, co = u % BM) --------
  unsigned va[A_U];
#pragma unroll
  for (int i = 0; i < A_U; ++i) {
    const int u = t + i * 256;
    const int kb = u / BM, co = m0 + u % BM;
    va[i] = co < Cout ? (unsigned)(kb * Cout + co) * 16u : kOOB;
  }
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t rw = make_rsrc(a.wt, a.wt_bytes);

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int csteps = Cin / BK;  // host guarantees Cin % 32 == 0
  const int nsteps = ntaps * csteps;
  uintx4 a_st[NST][A_U];
  float b_st[NST][CPT];
  // Load cursor (wave-uniform scalars): tap (kh, kw) and first channel of the
  // next step to fetch.  Everything below is branch-free so that the unrolled
  // step loop stays ONE basic block and the compiler keeps counted vmcnt waits:
  // steps past the end are fetched through out-of-range offsets (zeros) and
  // multiply into nothing.
  int c_step = 0, c_kh = 0, c_kw = 0, c_ci0 = 0;
  auto load_next = [&](uintx4* ra, float* rb) {
    const bool live = c_step < nsteps;
    const int wtap = MODE == 1 ? (a.kh0 + 2 * c_kh) * KW + a.kw0 + 2 * c_kw
                               : c_kh * ntw + c_kw;
    const int hi = bh0 + c_kh, wi = bw0 + c_kw;
    const bool ok = live && hi >= 0 && hi < bHin && wi >= 0 && wi < bWin;
    const unsigned vb = ok ? (unsigned)(boff + hi * bWin + wi) * 4u : kOOB;
    const unsigned sa = (unsigned)((wtap * Kp8 + (c_ci0 >> 3)) * Cout) * 16u;
#pragma unroll
    for (int i = 0; i < A_U; ++i) ra[i] = buf_load16(rw, live ? va[i] : kOOB, sa);
    const unsigned prow = (unsigned)Pin * 4u;
    unsigned so = (unsigned)c_ci0 * prow;
#pragma unroll
    for (int e = 0; e < CPT; ++e) {
      rb[e] = buf_load(rx, vb, so);
      so += prow;
    }
    // advance: channels fastest, then kw, then kh (scalar selects, no branch)
    ++c_step;
    c_ci0 += BK;
    const bool wc = c_ci0 >= Cin;
    c_ci0 = wc ? 0 : c_ci0;
    c_kw += wc ? 1 : 0;
    const bool wk = c_kw >= ntw;
    c_kw = wk ? 0 : c_kw;
    c_kh += wk ? 1 : 0;
  };
  auto store_tile = [&](int buf, const uintx4* ra, const float* rb) {
#pragma unroll
    for (int i = 0; i < A_U; ++i) {
      const int u = t + i * 256;
      As[buf * KB * BM + u] = ra[i];  // u = kb * BM + co: the image order
    }
#pragma unroll
    for (int h = 0; h < CPT / 8; ++h) {
      floatx8 f;
#pragma unroll
      for (int e = 0; e < 8; ++e) f[e] = rb[h * 8 + e];
      const bf16x8 v = __builtin_convertvector(f, bf16x8);
      const int kb = cg * (CPT / 8) + h;
      Bs[buf * KB * BN + kb * BN + bp] = __builtin_bit_cast(uintx4, v);
    }
  };
  auto compute = [&](int buf) {
    const uintx4* ap = As + buf * KB * BM + wm * (BM / 2) + l31;
    const uintx4* bq = Bs + buf * KB * BN + wn * (BN / 2) + l31;
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      uintx4 af[TM], bf[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[i] = ap[(2 * s + lk) * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[j] = bq[(2 * s + lk) * BN + j * 32];
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
              __builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]),
              acc[i][j], 0, 0, 0);
    }
  };

#pragma unroll
  for (int u = 0; u < NST - 1; ++u) load_next(a_st[u], b_st[u]);
  // the trip count is rounded up to whole rings; the padding steps are zeros
  for (int base = 0; base < nsteps; base += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      constexpr int ahead = NST - 1;
      load_next(a_st[(u + ahead) % NST], b_st[(u + ahead) % NST]);
      // LDS buffer (u & 1) was last read by the MFMAs two steps ago, which every
      // wave finished before it passed the previous barrier (NST is even, so the
      // buffer parity of a stage is static)
      store_tile(u & 1, a_st[u], b_st[u]);
      __syncthreads();
      compute(u & 1);
    }
  }

  // ---- epilogue: direct stores, 32 consecutive positions per accumulator row
  const bool has_res = a.residual != nullptr;
  const bool res8 = a.res_c8 != nullptr;  // residual as a C8 image
  const bool has_y = a.y != nullptr;      // fp32 output optional (C8-only nets)
  const bool c8out = a.y_c8 != nullptr;
  const bool relu = a.relu != 0;
  const bool has_aff = a.scale != nullptr, has_bias = a.bias != nullptr;
  // Loop order (round 3): row group g = 4 accumulator rows (one C8 half-row) x
  // all TN column tiles; every load of a group -- affine, bias, fp32 / C8
  // residual -- is issued before the group's first store.  The pointers of ConvK
  // may alias as far as hipcc knows, so a residual load written between stores
  // stays behind the previous store: the loop this replaces made up to 16
  // dependent round trips per 32 x 32 tile, each waited for with vmcnt(0) (ISA
  // reading; the 256 -> 1024 expansion convs ran at 1.2 TB/s).
  size_t colbase[TN], c8base[TN];
  bool jok[TN];
  const int prow = MODE == 1 ? a.Pfull : a.Pout;
  // per-channel operands through descriptors: an absent one has extent 0 and
  // loads as 0 -- no flag-dependent branch inside a group's load phase
  const rsrc_t r_sc = make_rsrc(a.scale, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_sh = make_rsrc(a.shift, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_bi = make_rsrc(a.bias, has_bias ? (unsigned)Cout * 4u : 0u);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int jc = n0 + wn * (BN / 2) + j * 32 + l31;
    jok[j] = jc < a.J;
    const int n = jc / a.Pout;
    int p = jc - n * a.Pout;
    // C8 images (side output, residual) exist for MODE 0 only: compact p
    c8base[j] = ((size_t)n * (Cout >> 3) * a.Pout + p) * 16;
    if (MODE == 1 && jok[j]) {
      int l, hc, wc;
      locate_out(a.g, p, l, hc, wc);
      p = a.foff[l] + (2 * hc + a.ph) * a.fW[l] + 2 * wc + a.pw;
    }
    colbase[j] = (size_t)n * Cout * prow + p;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rbase = m0 + wm * (BM / 2) + i * 32 + 4 * lk;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row0 = rbase + 8 * g;  // this lane's four rows: row0 .. row0 + 3
      // byte offset of the lane's 8-byte half of its C8 row within an image
      const size_t c8row = (size_t)(row0 >> 3) * a.Pout * 16 + (row0 & 4) * 2;
      float sc[4], sh[4], bi[4], rv[TN][4];
      uintx2 rraw[TN];
      // group fence: the loads of this group stay behind the previous group's
      // stores and none of them is consumed before all are issued
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned ro = (unsigned)(row0 + e) * 4u;  // >= Cout / absent: zeros
        sc[e] = buf_load(r_sc, ro, 0);
        sh[e] = buf_load(r_sh, ro, 0);
        bi[e] = buf_load(r_bi, ro, 0);
      }
      if (has_res) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            rv[j][e] = (jok[j] && row0 + e < Cout)
                           ? a.residual[colbase[j] + (size_t)(row0 + e) * prow]
                           : 0.0f;
      }
      if (MODE == 0 && res8) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          rraw[j] = (jok[j] && row0 < Cout)
                        ? *(const uintx2*)((const char*)a.res_c8 + c8base[j] + c8row)
                        : uintx2{0u, 0u};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!jok[j]) continue;
        floatx4_t q;
        floatx4_t rq = {0.0f, 0.0f, 0.0f, 0.0f};
        if (MODE == 0 && res8)
          rq = __builtin_convertvector(__builtin_bit_cast(bf16x4, rraw[j]), floatx4_t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int row = row0 + e;
          float v = 0.0f;
          if (row < Cout) {
            v = acc[i][j][r] * (has_aff ? sc[e] : 1.0f) + (sh[e] + bi[e]);
            if (MODE == 0 && a.y_raw)
              a.y_raw[colbase[j] + (size_t)row * prow] = acc[i][j][r];
            if (has_res) v += rv[j][e];
            if (MODE == 0 && res8) v += rq[e];
            if (relu) v = fmaxf(v, 0.0f);
            if (has_y) a.y[colbase[j] + (size_t)row * prow] = v;
          }
          q[e] = v;
        }
        // side output: this lane's four channels = half of a 16-byte C8 row
        if (MODE == 0 && c8out && row0 < Cout)
          *(uintx2*)((char*)a.y_c8 + c8base[j] + c8row) =
              __builtin_bit_cast(uintx2, __builtin_convertvector(q, bf16x4));
      }
    }
  }
}

// ------------------------------- forward/dgrad, LDS-tiled, bf16 C8 input --
// PMC of the kernel above (profiles/r02_pmc_tile_bf16/): with fp32 activations
// every B element costs a 4-byte gather plus a share of a conversion -- 11
// non-MFMA instructions per MFMA, the matrix pipe 8.5 % busy, INSTRUCTION-ISSUE
// bound.  Here the activation operand arrives as the bf16 channel-blocked image
// (N, C/8, P, 8) written by ld_conv_to_c8: the eight channels of one position
// are 16 contiguous bytes, i.e. exactly one row of the LDS image
// [BK/8][BN][8] -- one 16-byte load + one ds_write_b128 per (position, 8
// channels), no conversion, tap shifts stay 16-byte aligned and padding is
// still the buffer descriptor's out-of-range zero.  Per 32-deep k-step and
// wave: 4 loads, 4 ds_write_b128, 8 fragment reads, 8 MFMAs.
// Everything else (A image, pipeline, epilogue, MODE 1 parity classes) is the
// kernel above.
// BK = 64 (round 3): half the barriers and fragment-read restarts per MFMA; the
// table's d field carries BK.
// SCH = 1 (round 3): the LDS image of step u + 1 is written AFTER step u's
// barrier, behind the first fragment reads and under step u's MFMAs, instead of
// before the barrier where every wave waits for its own writes to land first.
// One barrier per step still orders both hazards (everyone has finished reading
// the buffer being overwritten; the writes of step u are waited for -- lgkmcnt --
// before barrier u + 1).  The register ring then runs NST - 2 steps ahead of the
// LDS write, so SCH = 1 needs NST = 4.  The table's cap field carries SCH.
template <int BM, int BN, int MODE, int NST, int BK, int SCH>
__global__ __launch_bounds__(256, 2) void conv_tile_c8_kernel(ConvK a) {
  static_assert(NST % 2 == 0, "ring length must be even (static LDS parity)");
  static_assert(BK == 32 || BK == 64, "k-step depth");
  static_assert(SCH == 0 || NST >= 4, "write-after-barrier needs a 4-slot ring");
  constexpr int KB = BK / 8;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int A_U = BM * KB / 256;  // 16-byte units of A per thread
  constexpr int CG = 256 / BN;        // k-block groups across the block
  constexpr int B_U = KB / CG;        // 16-byte units of B per thread
  static_assert(A_U >= 1 && B_U >= 1, "tile shape");
  __shared__ __attribute__((aligned(16))) uintx4 lds[2 * KB * (BM + BN)];
  uintx4* As = lds;                   // [2][KB][BM]
  uintx4* Bs = lds + 2 * KB * BM;     // [2][KB][BN]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lk = lane >> 5;
  const int mtiles = (a.Cout + BM - 1) / BM;
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (tile % mtiles) * BM;
  const int n0 = (tile / mtiles) * BN;

  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int KW = __builtin_amdgcn_readfirstlane(a.KW);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Kp8 = __builtin_amdgcn_readfirstlane(a.Kpad);
  const int ntw = __builtin_amdgcn_readfirstlane(MODE == 1 ? a.ntw : a.KW);
  const int ntaps = __builtin_amdgcn_readfirstlane(
      MODE == 1 ? a.nth * a.ntw : a.KH * a.KW);

  // ---- this thread's B column (one spatial position), in 16-byte units -----
  const int bp = t % BN;
  const int cg = __builtin_amdgcn_readfirstlane(t / BN);
  int bHin = 0, bWin = 0, boff = 0, bh0 = 0, bw0 = 0;
  {
    const int jb = n0 + bp;
    if (jb < a.J) {
      const int n = jb / a.Pout, p = jb - n * a.Pout;
      int bl, bho, bwo;
      locate_out(a.g, p, bl, bho, bwo);
      bHin = a.g.lv[bl].Hin;
      bWin = a.g.lv[bl].Win;
      boff = n * (Cin >> 3) * Pin + a.g.lv[bl].off_in;
      if (MODE == 1) {
        bh0 = bho + a.ch0;
        bw0 = bwo + a.cw0;
      } else {
        bh0 = bho * a.g.stride - a.g.pad;
        bw0 = bwo * a.g.stride - a.g.pad;
      }
    }
  }
  unsigned va[A_U];
#pragma unroll
  for (int i = 0; i < A_U; ++i) {
    const int u = t + i * 256;
    const int kb = u / BM, co = m0 + u % BM;
    va[i] = co < Cout ? (unsigned)(kb * Cout + co) * 16u : kOOB;
  }
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t rw = make_rsrc(a.wt, a.wt_bytes);

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int csteps = Cin / BK;  // host guarantees Cin % BK == 0
  const int nsteps = ntaps * csteps;
  uintx4 a_st[NST][A_U], b_st[NST][B_U];
  int c_step = 0, c_kh = 0, c_kw = 0, c_ci0 = 0;
  auto load_next = [&](uintx4* ra, uintx4* rb) {
    const bool live = c_step < nsteps;
    const int wtap = MODE == 1 ? (a.kh0 + 2 * c_kh) * KW + a.kw0 + 2 * c_kw
                               : c_kh * ntw + c_kw;
    const int hi = bh0 + c_kh, wi = bw0 + c_kw;
    const bool ok = live && hi >= 0 && hi < bHin && wi >= 0 && wi < bWin;
    const unsigned vb = ok ? (unsigned)(boff + hi * bWin + wi) * 16u : kOOB;
    const unsigned sa = (unsigned)((wtap * Kp8 + (c_ci0 >> 3)) * Cout) * 16u;
#pragma unroll
    for (int i = 0; i < A_U; ++i) ra[i] = buf_load16(rw, live ? va[i] : kOOB, sa);
    const unsigned prow = (unsigned)Pin * 16u;
    unsigned so = (unsigned)((c_ci0 >> 3) + cg * B_U) * prow;
#pragma unroll
    for (int q = 0; q < B_U; ++q) {
      rb[q] = buf_load16(rx, vb, so);
      so += prow;
    }
    ++c_step;
    c_ci0 += BK;
    const bool wc = c_ci0 >= Cin;
    c_ci0 = wc ? 0 : c_ci0;
    c_kw += wc ? 1 : 0;
    const bool wk = c_kw >= ntw;
    c_kw = wk ? 0 : c_kw;
    c_kh += wk ? 1 : 0;
  };
  auto store_tile = [&](int buf, const uintx4* ra, const uintx4* rb) {
#pragma unroll
    for (int i = 0; i < A_U; ++i) As[buf * KB * BM + t + i * 256] = ra[i];
#pragma unroll
    for (int q = 0; q < B_U; ++q)
      Bs[buf * KB * BN + (cg * B_U + q) * BN + bp] = rb[q];
  };
  auto frags = [&](int buf, int s, uintx4* af, uintx4* bf) {
    const uintx4* ap = As + buf * KB * BM + wm * (BM / 2) + l31;
    const uintx4* bq = Bs + buf * KB * BN + wn * (BN / 2) + l31;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = ap[(2 * s + lk) * BM + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = bq[(2 * s + lk) * BN + j * 32];
  };
  auto mfmas = [&](const uintx4* af, const uintx4* bf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
            __builtin_bit_cast(bf16x8, af[i]), __builtin_bit_cast(bf16x8, bf[j]),
            acc[i][j], 0, 0, 0);
  };
  auto compute = [&](int buf) {
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      uintx4 af[TM], bf[TN];
      frags(buf, s, af, bf);
      mfmas(af, bf);
    }
  };

#pragma unroll
  for (int u = 0; u < NST - 1; ++u) load_next(a_st[u], b_st[u]);
  if constexpr (SCH == 0) {
    for (int base = 0; base < nsteps; base += NST) {
#pragma unroll
      for (int u = 0; u < NST; ++u) {
        constexpr int ahead = NST - 1;
        load_next(a_st[(u + ahead) % NST], b_st[(u + ahead) % NST]);
        store_tile(u & 1, a_st[u], b_st[u]);
        __syncthreads();
        compute(u & 1);
      }
    }
  } else {
    store_tile(0, a_st[0], b_st[0]);
    for (int base = 0; base < nsteps; base += NST) {
#pragma unroll
      for (int u = 0; u < NST; ++u) {
        constexpr int ahead = NST - 1;
        load_next(a_st[(u + ahead) % NST], b_st[(u + ahead) % NST]);
        __syncthreads();  // image u complete; nobody still reads buffer (u + 1) & 1
        uintx4 af[TM], bf[TN];
        frags(u & 1, 0, af, bf);
        store_tile((u + 1) & 1, a_st[(u + 1) % NST], b_st[(u + 1) % NST]);
        mfmas(af, bf);
#pragma unroll
        for (int s = 1; s < BK / 16; ++s) {
          frags(u & 1, s, af, bf);
          mfmas(af, bf);
        }
      }
    }
  }

  // ---- epilogue: direct stores, 32 consecutive positions per accumulator row
  const bool has_res = a.residual != nullptr;
  const bool res8 = a.res_c8 != nullptr;  // residual as a C8 image
  const bool has_y = a.y != nullptr;      // fp32 output optional (C8-only nets)
  const bool c8out = a.y_c8 != nullptr;
  const bool relu = a.relu != 0;
  const bool has_aff = a.scale != nullptr, has_bias = a.bias != nullptr;
  // Loop order (round 3): row group g = 4 accumulator rows (one C8 half-row) x
  // all TN column tiles; every load of a group -- affine, bias, fp32 / C8
  // residual -- is issued before the group's first store.  The pointers of ConvK
  // may alias as far as hipcc knows, so a residual load written between stores
  // stays behind the previous store: the loop this replaces made up to 16
  // dependent round trips per 32 x 32 tile, each waited for with vmcnt(0) (ISA
  // reading; the 256 -> 1024 expansion convs ran at 1.2 TB/s).
  size_t colbase[TN], c8base[TN];
  bool jok[TN];
  const int prow = MODE == 1 ? a.Pfull : a.Pout;
  // per-channel operands through descriptors: an absent one has extent 0 and
  // loads as 0 -- no flag-dependent branch inside a group's load phase
  const rsrc_t r_sc = make_rsrc(a.scale, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_sh = make_rsrc(a.shift, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_bi = make_rsrc(a.bias, has_bias ? (unsigned)Cout * 4u : 0u);
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int jc = n0 + wn * (BN / 2) + j * 32 + l31;
    jok[j] = jc < a.J;
    const int n = jc / a.Pout;
    int p = jc - n * a.Pout;
    // C8 images (side output, residual) exist for MODE 0 only: compact p
    c8base[j] = ((size_t)n * (Cout >> 3) * a.Pout + p) * 16;
    if (MODE == 1 && jok[j]) {
      int l, hc, wc;
      locate_out(a.g, p, l, hc, wc);
      p = a.foff[l] + (2 * hc + a.ph) * a.fW[l] + 2 * wc + a.pw;
    }
    colbase[j] = (size_t)n * Cout * prow + p;
  }
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rbase = m0 + wm * (BM / 2) + i * 32 + 4 * lk;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row0 = rbase + 8 * g;  // this lane's four rows: row0 .. row0 + 3
      // byte offset of the lane's 8-byte half of its C8 row within an image
      const size_t c8row = (size_t)(row0 >> 3) * a.Pout * 16 + (row0 & 4) * 2;
      float sc[4], sh[4], bi[4], rv[TN][4];
      uintx2 rraw[TN];
      // group fence: the loads of this group stay behind the previous group's
      // stores and none of them is consumed before all are issued
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned ro = (unsigned)(row0 + e) * 4u;  // >= Cout / absent: zeros
        sc[e] = buf_load(r_sc, ro, 0);
        sh[e] = buf_load(r_sh, ro, 0);
        bi[e] = buf_load(r_bi, ro, 0);
      }
      if (has_res) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e)
            rv[j][e] = (jok[j] && row0 + e < Cout)
                           ? a.residual[colbase[j] + (size_t)(row0 + e) * prow]
                           : 0.0f;
      }
      if (MODE == 0 && res8) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
          rraw[j] = (jok[j] && row0 < Cout)
                        ? *(const uintx2*)((const char*)a.res_c8 + c8base[j] + c8row)
                        : uintx2{0u, 0u};
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (!jok[j]) continue;
        floatx4_t q;
        floatx4_t rq = {0.0f, 0.0f, 0.0f, 0.0f};
        if (MODE == 0 && res8)
          rq = __builtin_convertvector(__builtin_bit_cast(bf16x4, rraw[j]), floatx4_t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g + e;
          const int row = row0 + e;
          float v = 0.0f;
          if (row < Cout) {
            v = acc[i][j][r] * (has_aff ? sc[e] : 1.0f) + (sh[e] + bi[e]);
            if (MODE == 0 && a.y_raw)
              a.y_raw[colbase[j] + (size_t)row * prow] = acc[i][j][r];
            if (has_res) v += rv[j][e];
            if (MODE == 0 && res8) v += rq[e];
            if (relu) v = fmaxf(v, 0.0f);
            if (has_y) a.y[colbase[j] + (size_t)row * prow] = v;
          }
          q[e] = v;
        }
        // side output: this lane's four channels = half of a 16-byte C8 row
        if (MODE == 0 && c8out && row0 < Cout)
          *(uintx2*)((char*)a.y_c8 + c8base[j] + c8row) =
              __builtin_bit_cast(uintx2, __builtin_convertvector(q, bf16x4));
        // the conv result before the affine as a C8 image (round 6: what the BN
        // backward reads for d(gamma); 2 instead of 4 bytes per element)
        if (MODE == 0 && a.raw_c8 && row0 < Cout) {
          const floatx4_t rw = {acc[i][j][4 * g], acc[i][j][4 * g + 1],
                                acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          *(uintx2*)((char*)a.raw_c8 + c8base[j] + c8row) =
              __builtin_bit_cast(uintx2, __builtin_convertvector(rw, bf16x4));
        }
      }
    }
  }
}

// fp32 (N, C, P) -> bf16 channel-blocked (N, C/8, P, 8), round to nearest even.
// A thread converts one (n, 8-channel block, position): eight row reads that are
// each coalesced across the wave, one 16-byte write.
__global__ __launch_bounds__(256) void to_c8_kernel(const float* __restrict__ x, int C8,
                                                    int P, uintx4* __restrict__ out) {
  const int p = blockIdx.x * 256 + threadIdx.x;
  if (p >= P) return;
  const size_t blk = (size_t)blockIdx.z * C8 + blockIdx.y;  // (n, c8)
  const float* src = x + blk * 8 * P + p;
  floatx8 f;
#pragma unroll
  for (int e = 0; e < 8; ++e) f[e] = src[(size_t)e * P];
  const bf16x8 v = __builtin_convertvector(f, bf16x8);
  out[blk * P + p] = __builtin_bit_cast(uintx4, v);
}

// ---------------------------------------------------- wgrad, wave-private --
// One wavefront per workgroup, a 64(co) x 64(ci) tile of one tap and one
// j-split, 32 positions per step.  Both operand tiles are loaded coalesced
// along j (lane = position), rounded to bf16 and written to the wave's PRIVATE
// LDS tile as [row][j] with an 80-byte row pitch; a fragment (8 consecutive j
// of one row) is then ONE 16-byte read, and sixteen lanes x 80 B hit sixteen
// disjoint groups of four banks (conflict-free ds_read_b128).  LDS operations
// of one wave execute in order: no s_barrier anywhere.
constexpr int WB_J = 32;             // positions per step
constexpr int WB_PITCH = 2 * WB_J + 16;  // bytes per LDS row
// ------------------------------------- wgrad, wave-private, C8 operands ----
// The kernel below (fp32 operands) spends ~200 instructions per 8 MFMAs: both
// tiles are gathered 4 bytes per lane along positions, converted and written to
// LDS 2 bytes at a time, and its loads run only one 256-cycle step ahead of a
// ~2 000-cycle memory latency (profiles/r02_layers_bf16_c8.csv: 100-240 TFLOP/s).
// With BOTH operands as bf16 channel-blocked images (dY and X as (N, C/8, P, 8),
// the images the forward / the backward producers already wrote):
//   * a tile row = one position = 64 channels = eight 16-byte C8 rows: 4 loads +
//     4 ds_write_b128 per lane, operand and 32-position step, no conversion;
//     a tap shift moves whole C8 rows (always aligned), padding is the buffer
//     descriptor's zero;
//   * the reduction index (position) is the LDS ROW, the MFMA wants it inside a
//     lane: ds_read_b64_tr_b16 does that transpose in the LDS read -- per
//     16-lane group it fetches a [4 positions][16 channels] block (each lane
//     passes the address of one 8-byte piece: row t >> 2, channels 4 (t & 3)..)
//     and hands lane i the four positions of channel i (measured on the MI355X:
//     tools/probe/run_tr_probe.py); two such reads = one bf16x8 fragment;
//   * the loads run NR - 1 steps ahead through a register ring (8 x 16 B per
//     step and lane).
// LDS row pitch 144 B: conflict-free ds_write_b128 (36-dword stride) and
// transpose reads (rows 0 / 36 / 8 / 44 dwords apart, 8 dwords wide).
constexpr int WC_J = 32;       // positions per step
constexpr int WC_PITCH = 144;  // bytes per LDS row (64 channels + 16 pad)

__device__ __forceinline__ unsigned long long lds_read_tr(unsigned addr) {
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr) : "memory");
  return v;
}

// the same read with an instruction offset (must fold to a constant)
__device__ __forceinline__ unsigned long long lds_read_tr_off(unsigned addr, int off) {
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(off) : "memory");
  return v;
}

template <int NR>
__global__ __launch_bounds__(64, 2) void conv_wgrad_c8_kernel(WgradK a) {
  constexpr int TB = 64;
  constexpr int U = WC_J * (TB / 8) / 64;  // 16-byte units per lane and operand (4)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * WC_J * WC_PITCH];
  __shared__ int s_geo[LD_MAX_LEVELS * 6];
  unsigned char* As = lds;                    // [WC_J][WC_PITCH]  dY rows = positions
  unsigned char* Bs = lds + WC_J * WC_PITCH;  // [WC_J][WC_PITCH]  X

  const int lane = threadIdx.x;
  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Pout = __builtin_amdgcn_readfirstlane(a.Pout);
  const int nlev = __builtin_amdgcn_readfirstlane(a.g.num_levels);
  const int stride = __builtin_amdgcn_readfirstlane(a.g.stride);
  const int pad = __builtin_amdgcn_readfirstlane(a.g.pad);
  const int mt = (Cout + TB - 1) / TB, nt = (Cin + TB - 1) / TB;
  const int ntaps = a.KH * a.KW;
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int ntile = b % nt;
  b /= nt;
  const int mtile = b % mt;
  b /= mt;
  const int tap = b % ntaps;
  const int split = b / ntaps;
  const int m0 = mtile * TB, c0 = ntile * TB;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int jbeg = split * a.jchunk;
  const int jend = min(a.J, jbeg + a.jchunk);

  if (lane < LD_MAX_LEVELS) {
    const ld_conv_level_t lv = a.g.lv[lane];
    s_geo[lane * 6 + 0] = lv.Hin;
    s_geo[lane * 6 + 1] = lv.Win;
    s_geo[lane * 6 + 2] = lv.Hout;
    s_geo[lane * 6 + 3] = lv.Wout;
    s_geo[lane * 6 + 4] = lv.off_in;
    s_geo[lane * 6 + 5] = lv.off_out;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  // load side: lane = (position kq of the step, C8 block parity); unit i of the
  // lane is C8 block (lane >> 5) + 2 i of the tile
  const int kq = lane & (WC_J - 1);
  const int cb = lane >> 5;
  const int Co8 = Cout >> 3, Ci8 = Cin >> 3;
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t ry = make_rsrc(a.dy, a.dy_bytes);

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  uintx4 a_st[NR][U], b_st[NR][U];
  auto load_tile = [&](int j0, uintx4* ra, uintx4* rb) {
    const int j = j0 + kq;
    unsigned vy = kOOB, vx = kOOB;
    if (j < jend) {
      const int n = j / Pout, p = j - n * Pout;
      int l = 0;
      for (int i = 1; i < nlev; ++i)
        if (p >= s_geo[i * 6 + 5]) l = i;
      const int Hin = s_geo[l * 6 + 0], Win = s_geo[l * 6 + 1];
      const int Wout = s_geo[l * 6 + 3];
      const int r = p - s_geo[l * 6 + 5];
      const int ho = r / Wout, wo = r - ho * Wout;
      const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
      vy = (unsigned)((n * Co8 + (m0 >> 3) + cb) * Pout + p) * 16u;
      if (hi >= 0 && hi < Hin && wi >= 0 && wi < Win)
        vx = (unsigned)((n * Ci8 + (c0 >> 3) + cb) * Pin + s_geo[l * 6 + 4] +
                        hi * Win + wi) * 16u;
    }
    // C8 blocks past the channel count (ragged last tile) read zeros
    unsigned da = 2u * (unsigned)Pout * 16u, db = 2u * (unsigned)Pin * 16u;
    asm volatile("" : "+s"(da), "+s"(db));
    int na = (Co8 - (m0 >> 3) - cb + 1) / 2, nb = (Ci8 - (c0 >> 3) - cb + 1) / 2;
    asm volatile("" : "+v"(na), "+v"(nb));
    unsigned sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < U; ++i) {
      ra[i] = buf_load16(ry, i < na ? vy : kOOB, sa);
      rb[i] = buf_load16(rx, i < nb ? vx : kOOB, sb);
      sa += da;
      sb += db;
    }
  };
  auto store_tile = [&](const uintx4* ra, const uintx4* rb) {
    unsigned char* ap = As + kq * WC_PITCH + cb * 16;
    unsigned char* bp = Bs + kq * WC_PITCH + cb * 16;
#pragma unroll
    for (int i = 0; i < U; ++i) {
      *(uintx4*)(ap + i * 32) = ra[i];
      *(uintx4*)(bp + i * 32) = rb[i];
    }
  };
  // fragment read: lane l = (group g = l >> 4, t = l & 15); MFMA column
  // l & 31 = 16 (g & 1) + t, k half g >> 1; this lane's address piece = row
  // 8 (g >> 1) + (t >> 2) (+ 4 for the second read, + 16 per k16 step), channels
  // 16 (g & 1) + 4 (t & 3) (+ 32 per column tile)
  const int g4 = lane >> 4, t16 = lane & 15;
  const unsigned frag_off = (unsigned)((8 * (g4 >> 1) + (t16 >> 2)) * WC_PITCH +
                                       (16 * (g4 & 1) + 4 * (t16 & 3)) * 2);
  const unsigned a_base = (unsigned)(size_t)As + frag_off;
  const unsigned b_base = (unsigned)(size_t)Bs + frag_off;
  // raw transpose reads; the data is only valid after the s_waitcnt below,
  // which takes every result register as an in/out operand so that the
  // compiler cannot schedule an MFMA in front of it (the loads are inline asm:
  // hipcc does not count them)
  auto frag_addr = [&](unsigned base, int s, int tile) -> unsigned {
    return base + (unsigned)(s * 16 * WC_PITCH + tile * 64);
  };

  // The ring loads are UNCONDITIONAL (positions past jend resolve to the
  // descriptor's out-of-range zero, no memory traffic): with `if (step + ahead <
  // nsteps)` around them hipcc merged the two paths' counters and waited
  // vmcnt(0) before every LDS write -- each step paid the full latency of the
  // loads it had just issued and the ring hid nothing (round-3 ISA reading).
  const int nsteps = (jend - jbeg + WC_J - 1) / WC_J;
#pragma unroll
  for (int u = 0; u < NR - 1; ++u) load_tile(jbeg + u * WC_J, a_st[u], b_st[u]);
  for (int base = 0; base < nsteps; base += NR) {
#pragma unroll
    for (int u = 0; u < NR; ++u) {
      const int step = base + u;
      if (step >= nsteps) break;
      constexpr int ahead = NR - 1;
      load_tile(jbeg + (step + ahead) * WC_J, a_st[(u + ahead) % NR],
                b_st[(u + ahead) % NR]);
      store_tile(a_st[u], b_st[u]);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      unsigned long long fa[2][2][2], fb[2][2][2];  // [k16 step][tile][half]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned aa = frag_addr(a_base, s2, i), ab = frag_addr(b_base, s2, i);
          fa[s2][i][0] = lds_read_tr(aa);
          fa[s2][i][1] = lds_read_tr(aa + 4 * WC_PITCH);
          fb[s2][i][0] = lds_read_tr(ab);
          fb[s2][i][1] = lds_read_tr(ab + 4 * WC_PITCH);
        }
      asm volatile("s_waitcnt lgkmcnt(0)"
                   : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]),
                     "+v"(fa[0][1][1]), "+v"(fa[1][0][0]), "+v"(fa[1][0][1]),
                     "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fb[0][0][0]),
                     "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]),
                     "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]),
                     "+v"(fb[1][1][1])
                   :
                   : "memory");
      bf16x8 af[2][2], bfr[2][2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[s2][i] = __builtin_bit_cast(bf16x8, fa[s2][i]);
          bfr[s2][i] = __builtin_bit_cast(bf16x8, fb[s2][i]);
        }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s2][i], bfr[s2][j],
                                                                acc[i][j], 0, 0, 0);
      // fragment reads done (lgkmcnt 0) before the next tile's ds_writes
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  // slab store: [split][tap][co][ci], ci fastest (= lane & 31)
  const int l31 = lane & 31, lk = lane >> 5;
  float* slab = a.slabs + ((size_t)split * ntaps + tap) * Cout * Cin;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ci = c0 + j * 32 + l31;
    if (ci >= Cin) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (co < Cout) slab[(size_t)co * Cin + ci] = acc[i][j][r];
      }
  }
}

// ------------------------------ wgrad, workgroup-tiled, C8 operands (round 3) --
// The wave-private kernel above gives every wavefront its own 64 x 64 tile: 32
// flops per operand byte, and the (co tile, ci tile, tap) waves of a split all
// walk the same positions at the same time -- the 256-channel head layer pulls
// 1.65 GB through L2 -> L1 per launch (11.5 TB/s for 145 us), ~3 500 cycles per
// 32-position step at two waves per SIMD with the matrix pipe ~15 % busy.  Here
// four wavefronts (2 x 2) share a 128(co) x 128(ci) tile: each staged 16-byte
// unit feeds two waves (half the requests, 4 loads + 4 ds_write_b128 per lane
// and step instead of 16), and the LDS image of step u + 1 is written after
// step u's barrier, behind its transpose reads and under its MFMAs.
//   LDS image per operand: [32 positions][128 channels] bf16 = 256-byte rows, no
// padding; the 16-byte unit `ub` of row r sits at unit ub ^ 4 (r & 3) ^ ((r >> 2)
// & 3): the 32 lanes a ds_read_b64_tr_b16 services together (4 rows x one
// aligned group of 4 units) then cover all 64 banks once, and the 8 rows of a
// ds_write_b128 lane group land on 4 distinct bank groups (2-way).
// Same bf16 operands and the same per-wave MFMA order as the wave-private
// kernel for equal split boundaries.
constexpr int WT_PITCH = 256;

__device__ __forceinline__ unsigned wt_swz(int row) {
  return (unsigned)((4 * (row & 3)) ^ ((row >> 2) & 3));
}

template <int NR>
__global__ __launch_bounds__(256, 2) void conv_wgrad_c8_tile_kernel(WgradK a) {
  static_assert(NR >= 3, "the ring runs NR - 2 steps ahead of the LDS write");
  constexpr int TB = 128;
  constexpr int U = 2;  // 16-byte units per thread and operand: 32 x 16 / 256
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * WC_J * WT_PITCH];
  __shared__ int s_geo[LD_MAX_LEVELS * 6];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Pout = __builtin_amdgcn_readfirstlane(a.Pout);
  const int nlev = __builtin_amdgcn_readfirstlane(a.g.num_levels);
  const int stride = __builtin_amdgcn_readfirstlane(a.g.stride);
  const int pad = __builtin_amdgcn_readfirstlane(a.g.pad);
  const int mt = (Cout + TB - 1) / TB, nt = (Cin + TB - 1) / TB;
  const int ntaps = a.KH * a.KW;
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int ntile = b % nt;
  b /= nt;
  const int mtile = b % mt;
  b /= mt;
  const int tap = b % ntaps;
  const int split = b / ntaps;
  const int m0 = mtile * TB, c0 = ntile * TB;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int jbeg = split * a.jchunk;
  const int jend = min(a.J, jbeg + a.jchunk);

  if (t < LD_MAX_LEVELS) {
    const ld_conv_level_t lv = a.g.lv[t];
    s_geo[t * 6 + 0] = lv.Hin;
    s_geo[t * 6 + 1] = lv.Win;
    s_geo[t * 6 + 2] = lv.Hout;
    s_geo[t * 6 + 3] = lv.Wout;
    s_geo[t * 6 + 4] = lv.off_in;
    s_geo[t * 6 + 5] = lv.off_out;
  }
  __syncthreads();

  // load side: lane = (position kq of the step, C8 block cb of the tile); unit
  // i of the thread is C8 block cb + 8 i
  const int kq = lane & (WC_J - 1);
  const int cb = 2 * wave + (lane >> 5);
  const int Co8 = Cout >> 3, Ci8 = Cin >> 3;
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t ry = make_rsrc(a.dy, a.dy_bytes);

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  uintx4 a_st[NR][U], b_st[NR][U];
  auto load_tile = [&](int j0, uintx4* ra, uintx4* rb) {
    const int j = j0 + kq;
    unsigned vy = kOOB, vx = kOOB;
    if (j < jend) {
      const int n = j / Pout, p = j - n * Pout;
      int l = 0;
      for (int i = 1; i < nlev; ++i)
        if (p >= s_geo[i * 6 + 5]) l = i;
      const int Hin = s_geo[l * 6 + 0], Win = s_geo[l * 6 + 1];
      const int Wout = s_geo[l * 6 + 3];
      const int r = p - s_geo[l * 6 + 5];
      const int ho = r / Wout, wo = r - ho * Wout;
      const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
      vy = (unsigned)((n * Co8 + (m0 >> 3) + cb) * Pout + p) * 16u;
      if (hi >= 0 && hi < Hin && wi >= 0 && wi < Win)
        vx = (unsigned)((n * Ci8 + (c0 >> 3) + cb) * Pin + s_geo[l * 6 + 4] +
                        hi * Win + wi) * 16u;
    }
    // C8 blocks past the channel count (ragged last tile) read zeros
    const unsigned da = 8u * (unsigned)Pout * 16u, db = 8u * (unsigned)Pin * 16u;
    const int na = (Co8 - (m0 >> 3) - cb + 7) / 8, nb = (Ci8 - (c0 >> 3) - cb + 7) / 8;
    unsigned sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < U; ++i) {
      ra[i] = buf_load16(ry, i < na ? vy : kOOB, sa);
      rb[i] = buf_load16(rx, i < nb ? vx : kOOB, sb);
      sa += da;
      sb += db;
    }
  };
  const unsigned st_off0 = (unsigned)(kq * WT_PITCH) + (((unsigned)cb ^ wt_swz(kq)) << 4);
  const unsigned st_off1 =
      (unsigned)(kq * WT_PITCH) + (((unsigned)(cb + 8) ^ wt_swz(kq)) << 4);
  auto store_tile = [&](int buf, const uintx4* ra, const uintx4* rb) {
    unsigned char* As = lds + buf * 2 * WC_J * WT_PITCH;
    unsigned char* Bs = As + WC_J * WT_PITCH;
    *(uintx4*)(As + st_off0) = ra[0];
    *(uintx4*)(As + st_off1) = ra[1];
    *(uintx4*)(Bs + st_off0) = rb[0];
    *(uintx4*)(Bs + st_off1) = rb[1];
  };
  // fragment read (the wave-private kernel's lane map): lane l = (g = l >> 4,
  // t = l & 15) passes the address of the 8-byte piece at row 8 (g >> 1) + (t >>
  // 2) (+ 4 for the second read, + 16 per k16 step), channels 16 (g & 1) +
  // 4 (t & 3) (+ 32 per column tile, + 64 for the wave's half of the tile)
  const int g4 = lane >> 4, t16 = lane & 15;
  const int frow = 8 * (g4 >> 1) + (t16 >> 2);
  const unsigned fub = (unsigned)(2 * (g4 & 1) + ((t16 & 3) >> 1));  // unit in its group of 4
  const unsigned fhalf = (unsigned)((t16 & 3) & 1) * 8u;
  // row & 3 = t16 >> 2 for every read of this lane; (row >> 2) & 3 = 2 (g4 >> 1)
  // + h (h = second read)
  const unsigned sw0 = wt_swz(frow), sw1 = wt_swz(frow + 4);
  // eight per-lane addresses (operand, column tile, first / second read); the LDS
  // buffer (16 KB) and the k16 step (16 rows) are instruction offsets
  auto frag_base = [&](int op, int half64, int tile, int h) -> unsigned {
    const unsigned ub = (unsigned)(8 * half64 + 4 * tile) + fub;
    const unsigned row = (unsigned)(frow + 4 * h);
    return (unsigned)(size_t)lds + (unsigned)(op * WC_J * WT_PITCH) + row * WT_PITCH +
           ((ub ^ (h ? sw1 : sw0)) << 4) + fhalf;
  };
  unsigned fad[2][2][2];  // [operand][tile][h]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      fad[0][i][h] = frag_base(0, wm, i, h);
      fad[1][i][h] = frag_base(1, wn, i, h);
    }

  const int nsteps = (jend - jbeg + WC_J - 1) / WC_J;
  // unconditional ring loads and LDS writes (past jend: zeros) -- see the
  // wave-private kernel: a branch around them costs the counted vmcnt waits
#pragma unroll
  for (int u = 0; u < NR - 1; ++u) load_tile(jbeg + u * WC_J, a_st[u], b_st[u]);
  store_tile(0, a_st[0], b_st[0]);
  constexpr int UN = NR % 2 == 0 ? NR : 2 * NR;  // ring slot and LDS parity both static
  for (int base = 0; base < nsteps; base += UN) {
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int step = base + u;
      if (step >= nsteps) break;
      constexpr int ahead = NR - 1;
      load_tile(jbeg + (step + ahead) * WC_J, a_st[(u + ahead) % NR],
                b_st[(u + ahead) % NR]);
      __syncthreads();  // image `step` complete; nobody still reads buffer (u + 1) & 1
      unsigned long long fa[2][2][2], fb[2][2][2];  // [k16 step][tile][half]
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          constexpr int kBuf = 2 * WC_J * WT_PITCH, kS2 = 16 * WT_PITCH;
          const int off = (u & 1) * kBuf + s2 * kS2;  // compile-time after unrolling
          fa[s2][i][0] = lds_read_tr_off(fad[0][i][0], off);
          fa[s2][i][1] = lds_read_tr_off(fad[0][i][1], off);
          fb[s2][i][0] = lds_read_tr_off(fad[1][i][0], off);
          fb[s2][i][1] = lds_read_tr_off(fad[1][i][1], off);
        }
      // image step + 1 (zeros past the end), behind the reads, under the MFMAs
      store_tile((u + 1) & 1, a_st[(u + 1) % NR], b_st[(u + 1) % NR]);
      // LDS operations complete in order: at most the 4 newest (the writes
      // just issued) outstanding = all 16 transpose reads have landed
      asm volatile("s_waitcnt lgkmcnt(4)"
                   : "+v"(fa[0][0][0]), "+v"(fa[0][0][1]), "+v"(fa[0][1][0]),
                     "+v"(fa[0][1][1]), "+v"(fa[1][0][0]), "+v"(fa[1][0][1]),
                     "+v"(fa[1][1][0]), "+v"(fa[1][1][1]), "+v"(fb[0][0][0]),
                     "+v"(fb[0][0][1]), "+v"(fb[0][1][0]), "+v"(fb[0][1][1]),
                     "+v"(fb[1][0][0]), "+v"(fb[1][0][1]), "+v"(fb[1][1][0]),
                     "+v"(fb[1][1][1])
                   :
                   : "memory");
      bf16x8 af[2][2], bfr[2][2];
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          af[s2][i] = __builtin_bit_cast(bf16x8, fa[s2][i]);
          bfr[s2][i] = __builtin_bit_cast(bf16x8, fb[s2][i]);
        }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s2][i], bfr[s2][j],
                                                                acc[i][j], 0, 0, 0);
    }
  }
  // slab store: [split][tap][co][ci], ci fastest (= lane & 31)
  const int l31 = lane & 31, lk = lane >> 5;
  float* slab = a.slabs + ((size_t)split * ntaps + tap) * Cout * Cin;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ci = c0 + wn * 64 + j * 32 + l31;
    if (ci >= Cin) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (co < Cout) slab[(size_t)co * Cin + ci] = acc[i][j][r];
      }
  }
}

// VY / VX: the dY / X tile is fetched four positions per lane (one 16-byte load,
// two v_cvt_pk, one 8-byte LDS write: 8 load instructions per operand and step
// instead of 32 + 32 conversions + 32 two-byte LDS writes).  Four consecutive
// positions of one channel row are contiguous whenever Pout % 4 == 0 (VY) and,
// for X, when input position == output position (1x1, stride 1, no padding:
// VX).  A 3x3 conv keeps the per-element X path (tap shifts, padding).  The LDS
// image and the MFMA order are those of the per-element kernel: same bits.
template <bool VY, bool VX>
__global__ __launch_bounds__(64, 2) void conv_wgrad_wave_bf16_kernel(WgradK a) {
  constexpr int TB = 64;
  constexpr int RG = 64 / WB_J;   // rows covered by one load instruction (2)
  constexpr int RPER = TB / RG;   // rows per lane and operand (32)
  constexpr int RG4 = 64 / (WB_J / 4);  // rows per 16-byte load instruction (8)
  constexpr int RPER4 = TB / RG4;       // 16-byte loads per lane and operand (8)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * TB * WB_PITCH];
  __shared__ int s_geo[LD_MAX_LEVELS * 6];
  unsigned char* As = lds;                  // [TB][WB_PITCH]  dY rows (co)
  unsigned char* Bs = lds + TB * WB_PITCH;  // [TB][WB_PITCH]  X rows (ci)

  const int lane = threadIdx.x;
  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Pout = __builtin_amdgcn_readfirstlane(a.Pout);
  const int nlev = __builtin_amdgcn_readfirstlane(a.g.num_levels);
  const int stride = __builtin_amdgcn_readfirstlane(a.g.stride);
  const int pad = __builtin_amdgcn_readfirstlane(a.g.pad);
  const int mt = (Cout + TB - 1) / TB, nt = (Cin + TB - 1) / TB;
  const int ntaps = a.KH * a.KW;
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int ntile = b % nt;
  b /= nt;
  const int mtile = b % mt;
  b /= mt;
  const int tap = b % ntaps;
  const int split = b / ntaps;
  const int m0 = mtile * TB, c0 = ntile * TB;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int jbeg = split * a.jchunk;
  const int jend = min(a.J, jbeg + a.jchunk);

  if (lane < LD_MAX_LEVELS) {
    const ld_conv_level_t lv = a.g.lv[lane];
    s_geo[lane * 6 + 0] = lv.Hin;
    s_geo[lane * 6 + 1] = lv.Win;
    s_geo[lane * 6 + 2] = lv.Hout;
    s_geo[lane * 6 + 3] = lv.Wout;
    s_geo[lane * 6 + 4] = lv.off_in;
    s_geo[lane * 6 + 5] = lv.off_out;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  const int kq = lane % WB_J;  // per-element path: this lane's j within a step
  const int r0 = lane / WB_J;  // ... and its first row; rows r0 + RG*i
  const int kq4 = lane % (WB_J / 4);  // 16-byte path: group of four positions
  const int r4 = lane / (WB_J / 4);   // ... first row; rows r4 + RG4*i

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t ry = make_rsrc(a.dy, a.dy_bytes);
  float a_st[VY ? 1 : RPER], b_st[VX ? 1 : RPER];
  floatx4_t a_v4[VY ? RPER4 : 1], b_v4[VX ? RPER4 : 1];
  const bool rows_full = (m0 + TB <= Cout) && (c0 + TB <= Cin);

  auto load_tile = [&](int j0) {
    unsigned vy = kOOB, vx = kOOB;
    if (!VY || !VX) {  // per-element geometry
      const int j = j0 + kq;
      if (j < jend) {
        const int n = j / Pout, p = j - n * Pout;
        int l = 0;
        for (int i = 1; i < nlev; ++i)
          if (p >= s_geo[i * 6 + 5]) l = i;
        const int Hin = s_geo[l * 6 + 0], Win = s_geo[l * 6 + 1];
        const int Wout = s_geo[l * 6 + 3];
        const int r = p - s_geo[l * 6 + 5];
        const int ho = r / Wout, wo = r - ho * Wout;
        const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
        vy = (unsigned)(n * Cout * Pout + (m0 + r0) * Pout + p) * 4u;
        if (hi >= 0 && hi < Hin && wi >= 0 && wi < Win)
          vx = (unsigned)(n * Cin * Pin + (c0 + r0) * Pin + s_geo[l * 6 + 4] +
                          hi * Win + wi) * 4u;
      }
    }
    unsigned vy4 = kOOB, vx4 = kOOB;
    if (VY || VX) {  // four consecutive positions of one image (Pout % 4 == 0)
      const int j = j0 + 4 * kq4;
      if (j < jend) {
        const int n = j / Pout, p = j - n * Pout;
        vy4 = (unsigned)(n * Cout * Pout + (m0 + r4) * Pout + p) * 4u;
        vx4 = (unsigned)(n * Cin * Pin + (c0 + r4) * Pin + p) * 4u;  // VX: Pin == Pout
      }
    }
    // row offsets advance in SGPRs inside the step (see conv_wgrad_wave_kernel)
    unsigned da = (unsigned)RG * Pout * 4u, db = (unsigned)RG * Pin * 4u;
    unsigned da4 = (unsigned)RG4 * Pout * 4u, db4 = (unsigned)RG4 * Pin * 4u;
    asm volatile("" : "+s"(da), "+s"(db), "+s"(da4), "+s"(db4));
    int na = (Cout - m0 - r0 + RG - 1) / RG, nb = (Cin - c0 - r0 + RG - 1) / RG;
    int na4 = (Cout - m0 - r4 + RG4 - 1) / RG4, nb4 = (Cin - c0 - r4 + RG4 - 1) / RG4;
    if (rows_full) na = nb = na4 = nb4 = 1 << 20;
    asm volatile("" : "+v"(na), "+v"(nb), "+v"(na4), "+v"(nb4));
    if (VY) {
      unsigned so = 0;
#pragma unroll
      for (int i = 0; i < RPER4; ++i) {
        a_v4[i] = __builtin_bit_cast(floatx4_t, buf_load16(ry, i < na4 ? vy4 : kOOB, so));
        so += da4;
      }
    } else {
      unsigned so = 0;
#pragma unroll
      for (int i = 0; i < RPER; ++i) {
        a_st[i] = buf_load(ry, i < na ? vy : kOOB, so);
        so += da;
      }
    }
    if (VX) {
      unsigned so = 0;
#pragma unroll
      for (int i = 0; i < RPER4; ++i) {
        b_v4[i] = __builtin_bit_cast(floatx4_t, buf_load16(rx, i < nb4 ? vx4 : kOOB, so));
        so += db4;
      }
    } else {
      unsigned so = 0;
#pragma unroll
      for (int i = 0; i < RPER; ++i) {
        b_st[i] = buf_load(rx, i < nb ? vx : kOOB, so);
        so += db;
      }
    }
  };
  auto store_tile = [&]() {
    if (VY) {
      unsigned char* ap = As + r4 * WB_PITCH + kq4 * 8;
#pragma unroll
      for (int i = 0; i < RPER4; ++i)
        *(uintx2*)(ap + RG4 * i * WB_PITCH) =
            __builtin_bit_cast(uintx2, __builtin_convertvector(a_v4[i], bf16x4));
    } else {
      __bf16* ap = (__bf16*)(As + r0 * WB_PITCH) + kq;
#pragma unroll
      for (int i = 0; i < RPER; ++i) ap[RG * i * (WB_PITCH / 2)] = (__bf16)a_st[i];
    }
    if (VX) {
      unsigned char* bp = Bs + r4 * WB_PITCH + kq4 * 8;
#pragma unroll
      for (int i = 0; i < RPER4; ++i)
        *(uintx2*)(bp + RG4 * i * WB_PITCH) =
            __builtin_bit_cast(uintx2, __builtin_convertvector(b_v4[i], bf16x4));
    } else {
      __bf16* bp = (__bf16*)(Bs + r0 * WB_PITCH) + kq;
#pragma unroll
      for (int i = 0; i < RPER; ++i) bp[RG * i * (WB_PITCH / 2)] = (__bf16)b_st[i];
    }
  };

  const int nsteps = (jend - jbeg + WB_J - 1) / WB_J;
  const int l31 = lane & 31, lk = lane >> 5;
  if (nsteps > 0) load_tile(jbeg);
  for (int step = 0; step < nsteps; ++step) {
    store_tile();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (step + 1 < nsteps) load_tile(jbeg + (step + 1) * WB_J);
    // fragment of row (32 i + l31), positions 16 s + 8 lk + 0..7
    const unsigned char* ap = As + l31 * WB_PITCH + 16 * lk;
    const unsigned char* bp = Bs + l31 * WB_PITCH + 16 * lk;
    bf16x8 af[2][2], bfr[2][2];  // [k16-step][tile]
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[s][i] = *(const bf16x8*)(ap + i * 32 * WB_PITCH + s * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bfr[s][j] = *(const bf16x8*)(bp + j * 32 * WB_PITCH + s * 32);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i], bfr[s][j],
                                                              acc[i][j], 0, 0, 0);
    // all fragment reads of this tile are issued before the next tile's
    // ds_writes (LDS executes one wave's operations in order)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  // slab store: [split][tap][co][ci], ci fastest (= lane & 31)
  float* slab = a.slabs + ((size_t)split * ntaps + tap) * Cout * Cin;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ci = c0 + j * 32 + l31;
    if (ci >= Cin) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (co < Cout) slab[(size_t)co * Cin + ci] = acc[i][j][r];
      }
  }
}

// ------------------------------------------------ wgrad, workgroup-tiled ---
// The wave-private kernel above moves 64 + 64 rows through LDS for every 8
// MFMAs of one wavefront: at bf16 MFMA speed it is bound by the load / convert /
// ds_write issue rate, not by the matrix core.  Here four wavefronts (2 x 2)
// share a 128(co) x 128(ci) tile, so each staged row feeds two waves, and two
// consecutive positions travel together: dY as ONE 8-byte load (rows are 8-byte
// aligned when Pout is even), X as two gathered loads, packed by
// v_cvt_pk_bf16_f32 into ONE ds_write_b32 -- 72 instructions per 8 MFMAs
// instead of ~208.  Same 80-byte LDS row pitch and 16-byte fragment reads,
// double-buffered with one barrier per 32-position step.
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256, 2) void conv_wgrad_tile_bf16_kernel(WgradK a) {
  constexpr int TB = 128;
  constexpr int RGRP = 16;           // row groups of 16 threads (j-pairs)
  constexpr int RPER = TB / RGRP;    // rows per thread and operand (8)
  __shared__ __attribute__((aligned(16))) unsigned char lds[2 * 2 * TB * WB_PITCH];
  __shared__ int s_geo[LD_MAX_LEVELS * 6];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Pout = __builtin_amdgcn_readfirstlane(a.Pout);
  const int nlev = __builtin_amdgcn_readfirstlane(a.g.num_levels);
  const int stride = __builtin_amdgcn_readfirstlane(a.g.stride);
  const int pad = __builtin_amdgcn_readfirstlane(a.g.pad);
  const int mt = (Cout + TB - 1) / TB, nt = (Cin + TB - 1) / TB;
  const int ntaps = a.KH * a.KW;
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int ntile = b % nt;
  b /= nt;
  const int mtile = b % mt;
  b /= mt;
  const int tap = b % ntaps;
  const int split = b / ntaps;
  const int m0 = mtile * TB, c0 = ntile * TB;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int jbeg = split * a.jchunk;
  const int jend = min(a.J, jbeg + a.jchunk);

  if (t < LD_MAX_LEVELS) {
    const ld_conv_level_t lv = a.g.lv[t];
    s_geo[t * 6 + 0] = lv.Hin;
    s_geo[t * 6 + 1] = lv.Win;
    s_geo[t * 6 + 2] = lv.Hout;
    s_geo[t * 6 + 3] = lv.Wout;
    s_geo[t * 6 + 4] = lv.off_in;
    s_geo[t * 6 + 5] = lv.off_out;
  }
  __syncthreads();

  const int kq2 = t % 16;   // j-pair within the step: j = 2 kq2, 2 kq2 + 1
  const int r0 = t / 16;    // first row; rows r0 + 16 i

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t ry = make_rsrc(a.dy, a.dy_bytes);
  floatx2 a_st[RPER];
  float b_st[RPER][2];

  // input offset of position j for this tap (kOOB in the padding / past jend)
  auto x_off = [&](int j) -> unsigned {
    if (j >= jend) return kOOB;
    const int n = j / Pout, p = j - n * Pout;
    int l = 0;
    for (int i = 1; i < nlev; ++i)
      if (p >= s_geo[i * 6 + 5]) l = i;
    const int Hin = s_geo[l * 6 + 0], Win = s_geo[l * 6 + 1];
    const int Wout = s_geo[l * 6 + 3];
    const int r = p - s_geo[l * 6 + 5];
    const int ho = r / Wout, wo = r - ho * Wout;
    const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
    if (hi < 0 || hi >= Hin || wi < 0 || wi >= Win) return kOOB;
    return (unsigned)(n * Cin * Pin + (c0 + r0) * Pin + s_geo[l * 6 + 4] + hi * Win +
                      wi) * 4u;
  };
  auto load_tile = [&](int j0) {
    const int j = j0 + 2 * kq2;
    // dY: both positions of the pair in one 8-byte load (host: Pout even, so a
    // pair never straddles an image and the address is 8-byte aligned)
    unsigned vy = kOOB;
    if (j < jend) {
      const int n = j / Pout, p = j - n * Pout;
      vy = (unsigned)(n * Cout * Pout + (m0 + r0) * Pout + p) * 4u;
    }
    const unsigned vx0 = x_off(j), vx1 = x_off(j + 1);
    const int na = (Cout - m0 - r0 + RGRP - 1) / RGRP;  // valid rows of this thread
    const int nb = (Cin - c0 - r0 + RGRP - 1) / RGRP;
    unsigned sa = 0, sb = 0;
    const unsigned da = (unsigned)RGRP * Pout * 4u, db = (unsigned)RGRP * Pin * 4u;
#pragma unroll
    for (int i = 0; i < RPER; ++i) {
      const auto v = __builtin_amdgcn_raw_buffer_load_b64(ry, i < na ? vy : kOOB, sa, 0);
      a_st[i] = __builtin_bit_cast(floatx2, v);
      b_st[i][0] = buf_load(rx, i < nb ? vx0 : kOOB, sb);
      b_st[i][1] = buf_load(rx, i < nb ? vx1 : kOOB, sb);
      sa += da;
      sb += db;
    }
  };
  auto store_tile = [&](int buf) {
    unsigned char* As = lds + buf * 2 * TB * WB_PITCH;
    unsigned char* Bs = As + TB * WB_PITCH;
#pragma unroll
    for (int i = 0; i < RPER; ++i) {
      const int row = r0 + RGRP * i;
      const bf16x2 pa = __builtin_convertvector(a_st[i], bf16x2);
      floatx2 fb;
      fb[0] = b_st[i][0];
      fb[1] = b_st[i][1];
      const bf16x2 pb = __builtin_convertvector(fb, bf16x2);
      *(bf16x2*)(As + row * WB_PITCH + 4 * kq2) = pa;
      *(bf16x2*)(Bs + row * WB_PITCH + 4 * kq2) = pb;
    }
  };
  const int l31 = lane & 31, lk = lane >> 5;
  auto compute = [&](int buf) {
    const unsigned char* As = lds + buf * 2 * TB * WB_PITCH;
    const unsigned char* Bs = As + TB * WB_PITCH;
    const unsigned char* ap = As + (wm * 64 + l31) * WB_PITCH + 16 * lk;
    const unsigned char* bp = Bs + (wn * 64 + l31) * WB_PITCH + 16 * lk;
    bf16x8 af[2][2], bfr[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[s][i] = *(const bf16x8*)(ap + i * 32 * WB_PITCH + s * 32);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        bfr[s][j] = *(const bf16x8*)(bp + j * 32 * WB_PITCH + s * 32);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s][i], bfr[s][j],
                                                              acc[i][j], 0, 0, 0);
  };

  const int nsteps = (jend - jbeg + WB_J - 1) / WB_J;
  if (nsteps > 0) {
    load_tile(jbeg);
    store_tile(0);
  }
  __syncthreads();
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    const bool more = step + 1 < nsteps;
    if (more) load_tile(jbeg + (step + 1) * WB_J);
    compute(cur);
    if (more) store_tile(cur ^ 1);
    __syncthreads();
  }
  // slab store: [split][tap][co][ci], ci fastest (= lane & 31)
  float* slab = a.slabs + ((size_t)split * ntaps + tap) * Cout * Cin;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int ci = c0 + wn * 64 + j * 32 + l31;
    if (ci >= Cin) continue;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (co < Cout) slab[(size_t)co * Cin + ci] = acc[i][j][r];
      }
  }
}

// ---- weight images ----------------------------------------------------------
// (Cout, Cin, KH, KW) fp32 -> fwd [tap][Cin16/8][Cout][8] bf16 and/or
// bwd [KH*KW-1-tap][Cout16/8][Cin][8] bf16 (k rows beyond the channel count 0).
__device__ __forceinline__ void weight_transform_bf16_at(
    const float* __restrict__ w, int Cout, int Cin, int ntaps, __bf16* __restrict__ wt_fwd,
    __bf16* __restrict__ wt_bwd, size_t i) {
  const int ci8 = (Cin + 15) / 16 * 2, co8 = (Cout + 15) / 16 * 2;
  if (wt_fwd && i < (size_t)ntaps * ci8 * Cout * 8) {
    const int e = (int)(i & 7);
    size_t q = i >> 3;
    const int co = (int)(q % Cout);
    q /= Cout;
    const int kb = (int)(q % ci8), tap = (int)(q / ci8);
    const int ci = kb * 8 + e;
    wt_fwd[i] = (__bf16)(ci < Cin ? w[((size_t)co * Cin + ci) * ntaps + tap] : 0.0f);
  }
  if (wt_bwd && i < (size_t)ntaps * co8 * Cin * 8) {
    const int e = (int)(i & 7);
    size_t q = i >> 3;
    const int ci = (int)(q % Cin);
    q /= Cin;
    const int kb = (int)(q % co8), tapf = (int)(q / co8);
    const int co = kb * 8 + e;
    wt_bwd[i] = (__bf16)(co < Cout
                             ? w[((size_t)co * Cin + ci) * ntaps + (ntaps - 1 - tapf)]
                             : 0.0f);
  }
}

__global__ void conv_weight_transform_bf16_kernel(const float* __restrict__ w, int Cout,
                                                  int Cin, int ntaps,
                                                  __bf16* __restrict__ wt_fwd,
                                                  __bf16* __restrict__ wt_bwd) {
  weight_transform_bf16_at(w, Cout, Cin, ntaps, wt_fwd, wt_bwd,
                           (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ __launch_bounds__(256) void conv_weight_transform_bf16_batch_kernel(
    const ld_wt_job_t* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  const ld_wt_job_t j = jobs[block_job[blockIdx.x]];
  weight_transform_bf16_at(j.w, j.Cout, j.Cin, j.ntaps, (__bf16*)j.wt_fwd,
                           (__bf16*)j.wt_bwd,
                           (size_t)(blockIdx.x - j.first_block) * 256 + threadIdx.x);
}

// ---- shape dispatch ---------------------------------------------------------
struct StreamCfg {
  int tm, tn, wvm, d, ks;
  int sch = 0;  // C8 tiled family: 1 = LDS image written after the barrier
};
// Ring depth is bounded by the 6-bit vmcnt counter: D * (TM + 8 * TN) loads are
// in flight per wavefront and must stay below 64 (2x2 tiles: D = 2, 36 loads).
#define LD_BF16_SHAPES(X)                                                          \
  X(2, 2, 2, 2, 1) X(2, 2, 1, 2, 1) X(2, 1, 2, 4, 1) X(1, 2, 2, 2, 1)              \
  X(1, 1, 2, 4, 1) X(1, 1, 1, 4, 1) X(1, 1, 4, 4, 1) X(2, 1, 4, 4, 1)              \
  X(1, 1, 1, 4, 4) X(2, 1, 1, 4, 4) X(1, 2, 1, 2, 4) X(2, 2, 1, 2, 4)              \
  X(2, 2, 2, 1, 1) X(1, 1, 2, 1, 1) X(1, 1, 1, 1, 4)
// LDS-tiled kernel shapes, marked by wvm = 0: {BM / 32, BN / 32, 0, BK, NST}
// (NST = register stages of the load ring; the ks field carries it)
#define LD_BF16_TILE_SHAPES(X)                                                     \
  X(128, 128, 2) X(128, 128, 4) X(64, 128, 4) X(128, 64, 4) X(128, 64, 2)
// C8-input tiled kernel shapes (family 2): (BM, BN, NST, BK, SCH)
#define LD_C8_TILE_SHAPES(X)                                                       \
  X(128, 128, 2, 32, 0) X(128, 128, 4, 32, 0) X(64, 128, 4, 32, 0)                 \
  X(128, 64, 4, 32, 0) X(64, 64, 4, 32, 0) X(64, 128, 2, 32, 0)                    \
  X(128, 64, 2, 32, 0) X(64, 64, 2, 32, 0) X(128, 256, 2, 32, 0)                   \
  X(64, 256, 2, 32, 0)                                                             \
  X(128, 128, 2, 64, 0) X(64, 128, 2, 64, 0) X(128, 64, 2, 64, 0)                  \
  X(64, 64, 2, 64, 0) X(64, 64, 4, 64, 0) X(64, 128, 4, 64, 0)                     \
  X(128, 128, 4, 32, 1) X(64, 128, 4, 32, 1) X(128, 64, 4, 32, 1)                  \
  X(64, 64, 4, 32, 1) X(64, 64, 4, 64, 1) X(64, 128, 4, 64, 1)
// ks = 8: the 8-wave LDS-DMA kernel of conv_t256.hip (BM x BN workgroup tiles,
// 64-deep steps, two LDS stages)
#define LD_C8_T256_SHAPES(X) X(256, 256) X(256, 192) X(128, 256) X(256, 128)
constexpr StreamCfg kC8Cfgs[] = {
#define LD_ROW(BM_, BN_, NST_, BK_, SCH_) {BM_ / 32, BN_ / 32, 0, BK_, NST_, SCH_},
    LD_C8_TILE_SHAPES(LD_ROW)
#undef LD_ROW
#define LD_ROW(BM_, BN_) {BM_ / 32, BN_ / 32, 0, 64, 8, 0},
    LD_C8_T256_SHAPES(LD_ROW)
#undef LD_ROW
};
constexpr int kNumC8Cfgs = sizeof(kC8Cfgs) / sizeof(kC8Cfgs[0]);
constexpr StreamCfg kCfgs[] = {
#define LD_ROW(TM_, TN_, WVM_, D_, KS_) {TM_, TN_, WVM_, D_, KS_},
    LD_BF16_SHAPES(LD_ROW)
#undef LD_ROW
#define LD_ROW(BM_, BN_, NST_) {BM_ / 32, BN_ / 32, 0, 32, NST_},
    LD_BF16_TILE_SHAPES(LD_ROW)
#undef LD_ROW
};
constexpr int kNumCfgs = sizeof(kCfgs) / sizeof(kCfgs[0]);

inline int mode_taps(const ConvK& k) {
  return k.nth > 0 ? k.nth * k.ntw : k.KH * k.KW;
}

template <int MODE>
int launch_c8_cfg(const ConvK& k, const StreamCfg& c, hipStream_t stream) {
  const int BM = c.tm * 32, BN = c.tn * 32;
  if (c.d < 32 || k.Cin % c.d != 0) return LD_EUNSUPPORTED;
  if (c.ks == 8) return ld_bf16_t256_launch(MODE, k, BM, BN, stream);
  const int nb = ((k.Cout + BM - 1) / BM) * ((k.J + BN - 1) / BN);
#define LD_CASE(BM_, BN_, NST_, BK_, SCH_)                                         \
  if (BM == BM_ && BN == BN_ && c.ks == NST_ && c.d == BK_ && c.sch == SCH_) {     \
    LD_LAUNCH((conv_tile_c8_kernel<BM_, BN_, MODE, NST_, BK_, SCH_>),     \
                       dim3(nb), dim3(256), 0, stream, k);                         \
    return (int)hipGetLastError();                                                 \
  }
  LD_C8_TILE_SHAPES(LD_CASE)
#undef LD_CASE
  return LD_EUNSUPPORTED;
}

inline bool c8_cfg_fits(const ConvK& k, const StreamCfg& c) {
  if (c.d < 32 || k.Cin % c.d != 0) return false;
  if (c.ks == 8) return ld_bf16_t256_fits(k, c.tm * 32, c.tn * 32);
  const int BM = c.tm * 32;
  const int cout32 = (k.Cout + 31) / 32 * 32;
  return BM <= cout32 + 32;
}

// model pick: rounds over the 256 CUs (2 resident workgroups each), larger
// tiles re-use operands better
inline int c8_cfg_model(const ConvK& k) {
  int best = -1;
  double best_t = 0;
  for (int i = 0; i < kNumC8Cfgs; ++i) {
    const StreamCfg& c = kC8Cfgs[i];
    // 64-deep steps and the write-after-barrier schedule: by table only
    if (c.d != 32 || c.sch != 0 || !c8_cfg_fits(k, c)) continue;
    const long nb = (long)((k.Cout + c.tm * 32 - 1) / (c.tm * 32)) *
                    ((k.J + c.tn * 32 - 1) / (c.tn * 32));
    const double rounds = nb <= 512 ? (double)((nb + 255) / 256) : (double)nb / 256.0;
    const int area = c.tm * c.tn;
    const double eff = area >= 16 ? 1.0 : area >= 8 ? 0.8 : 0.6;
    const double t = rounds * area / eff / (c.ks >= 4 ? 1.05 : 1.0);
    if (best < 0 || t < best_t) {
      best = i;
      best_t = t;
    }
  }
  return best;
}

template <int MODE>
int launch_cfg(const ConvK& k, const StreamCfg& c, hipStream_t stream) {
  if (c.wvm == 0) {  // LDS-tiled kernel
    const int BM = c.tm * 32, BN = c.tn * 32;
    if (k.Cin % 32 != 0) return LD_EUNSUPPORTED;
    const int nb = ((k.Cout + BM - 1) / BM) * ((k.J + BN - 1) / BN);
#define LD_CASE(BM_, BN_, NST_)                                                    \
  if (BM == BM_ && BN == BN_ && c.ks == NST_) {                                    \
    LD_LAUNCH((conv_tile_bf16_kernel<BM_, BN_, MODE, NST_>), dim3(nb),    \
                       dim3(256), 0, stream, k);                                   \
    return (int)hipGetLastError();                                                 \
  }
    LD_BF16_TILE_SHAPES(LD_CASE)
#undef LD_CASE
    return LD_EUNSUPPORTED;
  }
  const int bm = (c.ks == 4 ? 1 : c.wvm) * c.tm * 32;
  const int bn = (c.ks == 4 ? 1 : 4 / c.wvm) * c.tn * 32;
  const int nb = ((k.Cout + bm - 1) / bm) * ((k.J + bn - 1) / bn);
#define LD_CASE(TM_, TN_, WVM_, D_, KS_)                                           \
  if (c.tm == TM_ && c.tn == TN_ && c.wvm == WVM_ && c.d == D_ && c.ks == KS_) {   \
    LD_LAUNCH(                                                            \
        (conv_stream_bf16_kernel<TM_, TN_, WVM_, MODE, D_,                         \
                                 (KS_ == 4 ? (TM_ * TN_ < 2 ? 4 : TM_ * TN_ < 4 ? 3 : 2) : 2), KS_>),       \
        dim3(nb), dim3(256), 0, stream, k);                                        \
    return (int)hipGetLastError();                                                 \
  }
  LD_BF16_SHAPES(LD_CASE)
#undef LD_CASE
  return LD_EUNSUPPORTED;
}

inline bool cfg_fits(const ConvK& k, const StreamCfg& c) {
  if (c.wvm == 0) {
    if (k.Cin % 32 != 0) return false;
    const int BM = c.tm * 32;
    const int cout32 = (k.Cout + 31) / 32 * 32;
    return BM <= cout32 + 32;  // at most one 32-row slab of padding
  }
  const int steps = k.Cin / 16;
  if (steps % c.d != 0) return false;
  if (c.d == 1 && steps % 2 == 0) return false;  // a deeper ring covers it
  const int bm = c.wvm * c.tm * 32;
  const int cout32 = (k.Cout + 31) / 32 * 32;
  if (c.ks == 4) {
    const int nchunks = mode_taps(k) * (steps / c.d);
    if (nchunks < 16) return false;
  }
  if (c.wvm > 1 && bm > cout32) return false;  // whole waves of padding rows
  if (c.wvm == 1 && c.tm * 32 >= cout32 + 32) return false;
  return true;
}

// model-based pick: workgroup rounds over 256 CUs x per-shape efficiency
inline int cfg_model(const ConvK& k) {
  int best = -1;
  double best_t = 0;
  for (int i = 0; i < kNumCfgs; ++i) {
    const StreamCfg& c = kCfgs[i];
    if (!cfg_fits(k, c)) continue;
    if (c.wvm == 0) {  // LDS-tiled: ~2x the streaming kernel's pipe efficiency
      const long nb = (long)((k.Cout + c.tm * 32 - 1) / (c.tm * 32)) *
                      ((k.J + c.tn * 32 - 1) / (c.tn * 32));
      const double rounds =
          nb <= 512 ? (double)((nb + 255) / 256) : (double)nb / 256.0;
      const double t = rounds * c.tm * c.tn / (c.ks >= 4 ? 1.6 : 1.2);
      if (best < 0 || t < best_t) {
        best = i;
        best_t = t;
      }
      continue;
    }
    const int bm = (c.ks == 4 ? 1 : c.wvm) * c.tm * 32;
    const int bn = (c.ks == 4 ? 1 : 4 / c.wvm) * c.tn * 32;
    const long nb = (long)((k.Cout + bm - 1) / bm) * ((k.J + bn - 1) / bn);
    const int area = c.tm * c.tn;
    // operand reuse matters more than at fp32 rates: the loads per MFMA of a
    // 1x1 wave tile are twice those of a 2x2 one
    const double eff = (area >= 4 ? 0.90 : area >= 2 ? 0.70 : 0.50) *
                       (c.ks == 4 ? 0.95 : 1.0) * (c.d >= 2 ? 1.0 : 0.6);
    const int occ = area >= 4 ? 2 : 4;  // resident workgroups per CU
    const double rounds =
        nb <= 256L * occ ? (double)((nb + 255) / 256) : (double)nb / 256.0;
    const double t = rounds * (c.ks == 4 ? 1 : 4) * area / eff;
    if (best < 0 || t < best_t) {
      best = i;
      best_t = t;
    }
  }
  return best;
}

inline int cfg_index(const LdTuneCfg& c) {
  for (int i = 0; i < kNumCfgs; ++i)
    if (kCfgs[i].tm == c.tm && kCfgs[i].tn == c.tn && kCfgs[i].wvm == c.wvm &&
        kCfgs[i].d == c.d && kCfgs[i].ks == c.ks)
      return i;
  return -1;
}

inline int c8_cfg_index(const LdTuneCfg& c) {
  for (int i = 0; i < kNumC8Cfgs; ++i)
    if (kC8Cfgs[i].tm == c.tm && kC8Cfgs[i].tn == c.tn && kC8Cfgs[i].ks == c.ks &&
        kC8Cfgs[i].d == c.d && kC8Cfgs[i].sch == c.cap && c.wvm == 0)
      return i;
  return -1;
}

template <int MODE>
int launch_c8(const ConvK& k, hipStream_t stream) {
  if (k.Cin % 32 != 0) return LD_EUNSUPPORTED;
  // "4x4x2", "4x4x2x64" or "4x4x4x32x1": BM/32 x BN/32 x NST [x BK [x SCH]]
  if (const char* env = getenv("LD_CONV_C8_SHAPE")) {
    StreamCfg c{0, 0, 0, 32, 0, 0};
    if (sscanf(env, "%dx%dx%dx%dx%d", &c.tm, &c.tn, &c.ks, &c.d, &c.sch) >= 3 &&
        c8_cfg_fits(k, c)) {
      const int rc = launch_c8_cfg<MODE>(k, c, stream);
      if (rc != LD_EUNSUPPORTED) return rc;
    }
  }
  int pick = -1;
  LdTuneCfg t;
  if (ld_tune_lookup(make_tune_key(MODE, 2, k), &t)) {
    pick = c8_cfg_index(t);
    if (pick >= 0 && !c8_cfg_fits(k, kC8Cfgs[pick])) pick = -1;
  }
  if (pick < 0) pick = c8_cfg_model(k);
  if (pick < 0) return LD_EUNSUPPORTED;
  return launch_c8_cfg<MODE>(k, kC8Cfgs[pick], stream);
}

template <int MODE>
int tune_c8(const ConvK& k, hipStream_t stream) {
  if (k.Cin % 32 != 0) return 1;
  const LdTuneKey key = make_tune_key(MODE, 2, k);
  LdTuneCfg have;
  if (ld_tune_lookup(key, &have)) return 1;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cap);
  if (cap != hipStreamCaptureStatusNone) return LD_EUNSUPPORTED;
  int pick = c8_cfg_model(k);
  if (pick < 0) return 1;
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  constexpr int kReps = 3;
  float best_ms = -1.0f;
  for (int i = 0; i < kNumC8Cfgs; ++i) {
    if (!c8_cfg_fits(k, kC8Cfgs[i])) continue;
    if (launch_c8_cfg<MODE>(k, kC8Cfgs[i], stream) != 0) continue;
    float ms = -1.0f;
    for (int trial = 0; trial < 2; ++trial) {
      (void)hipEventRecord(e0, stream);
      for (int rep = 0; rep < kReps; ++rep) launch_c8_cfg<MODE>(k, kC8Cfgs[i], stream);
      (void)hipEventRecord(e1, stream);
      if (hipEventSynchronize(e1) != hipSuccess) break;
      float t = 0.0f;
      (void)hipEventElapsedTime(&t, e0, e1);
      if (ms < 0.0f || t < ms) ms = t;
    }
    if (ms < 0.0f) continue;
    if (best_ms < 0.0f || ms < best_ms) {
      best_ms = ms;
      pick = i;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const StreamCfg& c = kC8Cfgs[pick];
  if (const char* lg = getenv("LD_CONV_TUNE_LOG"))
    if (lg[0] == '1') {
      const double fl = 2.0 * k.J * k.Cout * k.Cin * mode_taps(k);
      fprintf(stderr,
              "[ld_conv c8] mode %d Cin %d Cout %d k %dx%d s%d J %d lv %d -> "
              "%dx%dx%d bk%d sch%d  %.1f TFLOP/s\n",
              MODE, k.Cin, k.Cout, k.KH, k.KW, k.g.stride, k.J, k.g.num_levels,
              c.tm * 32, c.tn * 32, c.ks, c.d, c.sch,
              best_ms > 0 ? fl / (best_ms * 1e-3 / kReps) / 1e12 : 0.0);
    }
  if (best_ms > 0.0f) ld_tune_store(key, LdTuneCfg{c.tm, c.tn, 0, c.d, c.ks, c.sch});
  return 0;
}

template <int MODE>
int launch_bf16(const ConvK& k, hipStream_t stream) {
  if (k.x_c8) return launch_c8<MODE>(k, stream);
  if (k.Cin % 16 != 0) return LD_EUNSUPPORTED;
  if (const char* env = getenv("LD_CONV_BF16_SHAPE")) {  // "2x2x2x4x1": force a shape
    StreamCfg c;
    if (sscanf(env, "%dx%dx%dx%dx%d", &c.tm, &c.tn, &c.wvm, &c.d, &c.ks) == 5 &&
        (c.wvm == 0 || (k.Cin / 16) % c.d == 0)) {
      const int rc = launch_cfg<MODE>(k, c, stream);
      if (rc != LD_EUNSUPPORTED) return rc;
    }
  }
  int pick = -1;
  LdTuneCfg t;
  if (ld_tune_lookup(make_tune_key(MODE, 1, k), &t)) {
    pick = cfg_index(t);
    if (pick >= 0 && !cfg_fits(k, kCfgs[pick])) pick = -1;
  }
  if (pick < 0) pick = cfg_model(k);
  if (pick < 0) return LD_EUNSUPPORTED;
  return launch_cfg<MODE>(k, kCfgs[pick], stream);
}

template <int MODE>
int tune_bf16(const ConvK& k, hipStream_t stream) {
  if (k.x_c8) return tune_c8<MODE>(k, stream);
  if (k.Cin % 16 != 0) return 1;
  const LdTuneKey key = make_tune_key(MODE, 1, k);
  LdTuneCfg have;
  if (ld_tune_lookup(key, &have)) return 1;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cap);
  if (cap != hipStreamCaptureStatusNone) return LD_EUNSUPPORTED;
  int pick = cfg_model(k);
  if (pick < 0) return 1;
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  constexpr int kReps = 3;
  float best_ms = -1.0f;
  for (int i = 0; i < kNumCfgs; ++i) {
    if (!cfg_fits(k, kCfgs[i])) continue;
    if (launch_cfg<MODE>(k, kCfgs[i], stream) != 0) continue;
    float ms = -1.0f;
    for (int trial = 0; trial < 2; ++trial) {
      (void)hipEventRecord(e0, stream);
      for (int rep = 0; rep < kReps; ++rep) launch_cfg<MODE>(k, kCfgs[i], stream);
      (void)hipEventRecord(e1, stream);
      if (hipEventSynchronize(e1) != hipSuccess) break;
      float t = 0.0f;
      (void)hipEventElapsedTime(&t, e0, e1);
      if (ms < 0.0f || t < ms) ms = t;
    }
    if (ms < 0.0f) continue;
    if (best_ms < 0.0f || ms < best_ms) {
      best_ms = ms;
      pick = i;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const StreamCfg& c = kCfgs[pick];
  if (const char* lg = getenv("LD_CONV_TUNE_LOG"))
    if (lg[0] == '1') {
      const double fl = 2.0 * k.J * k.Cout * k.Cin * mode_taps(k);
      fprintf(stderr,
              "[ld_conv bf16] mode %d Cin %d Cout %d k %dx%d s%d J %d lv %d -> "
              "%dx%dx%dx%dx%d  %.1f TFLOP/s\n",
              MODE, k.Cin, k.Cout, k.KH, k.KW, k.g.stride, k.J, k.g.num_levels, c.tm,
              c.tn, c.wvm, c.d, c.ks,
              best_ms > 0 ? fl / (best_ms * 1e-3 / kReps) / 1e12 : 0.0);
    }
  if (best_ms > 0.0f) ld_tune_store(key, LdTuneCfg{c.tm, c.tn, c.wvm, c.d, c.ks});
  return 0;
}

}  // namespace

int ld_bf16_stream_launch(int mode, const ConvK& k, hipStream_t stream) {
  return mode == 1 ? launch_bf16<1>(k, stream) : launch_bf16<0>(k, stream);
}

int ld_bf16_stream_tune(int mode, const ConvK& k, hipStream_t stream) {
  return mode == 1 ? tune_bf16<1>(k, stream) : tune_bf16<0>(k, stream);
}

// Which wgrad kernel a geometry gets: the workgroup-tiled one needs even Pout
// (8-byte dY pairs) and at least one full 128-row tile per operand.
bool ld_bf16_wgrad_tiled(int Cout, int Cin, int Pout) {
  // Measured on the C2 step (profiles/r02_bench_s1.json vs r02_bench_s4.json):
  // the wave-private kernel runs the step's weight gradients at 154-160
  // TFLOP/s, the workgroup-tiled one at 128 -- its barrier per 32 positions
  // costs more than the halved staging work saves while the kernel is
  // issue-bound.  The tiled kernel therefore is opt-in
  // (LD_CONV_BF16_WGRAD=tile) until the activations are stored in bf16.
  const char* env = getenv("LD_CONV_BF16_WGRAD");
  if (!env || env[0] != 't') return false;
  return Pout % 2 == 0 && Cout >= 128 && Cin >= 128;
}

int ld_bf16_wgrad_splits(int Cout, int Cin, int ntaps, int J) {
  // ~3 resident workgroups per CU; at least 8 steps of 32 positions per split
  const int tiles = ((Cout + 127) / 128) * ((Cin + 127) / 128) * ntaps;
  int sp = (768 + tiles - 1) / tiles;
  const int max_by_k = (J + 8 * WB_J - 1) / (8 * WB_J);
  sp = max(1, min(min(sp, max_by_k), 256));
  const size_t wbytes = (size_t)ntaps * Cout * Cin * sizeof(float);
  while (sp > 1 && sp * wbytes > ((size_t)256 << 20)) --sp;  // slab budget
  // no empty trailing split
  for (;;) {
    int jc = (J + sp - 1) / sp;
    jc = (jc + WB_J - 1) / WB_J * WB_J;
    if (sp == 1 || (long)(sp - 1) * jc < J) break;
    --sp;
  }
  return sp;
}

// C8 weight gradient: the workgroup-tiled kernel (128 x 128 tiles) whenever both
// channel counts fill a tile; LD_CONV_WGRAD_C8_KERNEL=wave / tile forces one.
bool ld_bf16_wgrad_c8_tiled(int Cout, int Cin) {
  if (const char* env = getenv("LD_CONV_WGRAD_C8_KERNEL")) {
    if (env[0] == 'w') return false;
    if (env[0] == 't') return true;
  }
  return Cout >= 128 && Cin >= 128;
}

// j-split count of the tiled kernel: whole rounds of the 2 workgroups per CU its
// registers allow, at least 8 steps of 32 positions per split
int ld_bf16_wgrad_c8_tile_splits(int Cout, int Cin, int ntaps, int J) {
  const int tiles = ((Cout + 127) / 128) * ((Cin + 127) / 128) * ntaps;
  const int slots = 512;     // two workgroups per CU (190 VGPRs)
  // prologue + slab store + this split's share of the reduce pass (64 KB written,
  // 64 KB read back per workgroup), in steps of 32 positions; LD_WGRAD_C8_FIXED
  // overrides (round 5 sweep: tools/sessions/r05_s9.sh)
  static const double fixed = [] {
    const char* e = getenv("LD_WGRAD_C8_FIXED");
    const double v = e ? atof(e) : 0.0;
    return v > 0.0 ? v : 6.0;
  }();
  const size_t wbytes = (size_t)ntaps * Cout * Cin * sizeof(float);
  int best = 1;
  double best_cost = 0;
  for (int sp = 1; sp <= 128; ++sp) {
    if (sp > 1 && sp * wbytes > ((size_t)256 << 20)) break;  // slab budget
    int jchunk = (J + sp - 1) / sp;
    jchunk = (jchunk + WC_J - 1) / WC_J * WC_J;
    if (sp > 1 && (long)(sp - 1) * jchunk >= J) continue;  // empty last split
    const int steps = jchunk / WC_J;
    if (sp > 1 && steps < 8) break;
    const long blocks = (long)tiles * sp;
    const double rounds = (double)((blocks + slots - 1) / slots);
    const double cost = rounds * (steps + fixed) + 1e-3 * sp;
    if (sp == 1 || cost < best_cost) {
      best = sp;
      best_cost = cost;
    }
  }
  return best;
}

// both operands as C8 images (k.x / k.dy point at them; extents in bytes)
int ld_bf16_wgrad_c8_launch(const WgradK& k, hipStream_t stream) {
  if (k.Cin % 8 != 0 || k.Cout % 8 != 0) return LD_EUNSUPPORTED;
  const int ntaps = k.KH * k.KW;
  if (ld_bf16_wgrad_c8_tiled(k.Cout, k.Cin)) {
    const int blocks =
        ((k.Cout + 127) / 128) * ((k.Cin + 127) / 128) * ntaps * k.splits;
    LD_LAUNCH(conv_wgrad_c8_tile_kernel<4>, dim3(blocks), dim3(256), 0, stream,
                       k);
    return (int)hipGetLastError();
  }
  const int blocks = ((k.Cout + 63) / 64) * ((k.Cin + 63) / 64) * ntaps * k.splits;
  const char* env = getenv("LD_CONV_WGRAD_C8_RING");
  if (env && env[0] == '2')
    LD_LAUNCH(conv_wgrad_c8_kernel<2>, dim3(blocks), dim3(64), 0, stream, k);
  else
    LD_LAUNCH(conv_wgrad_c8_kernel<3>, dim3(blocks), dim3(64), 0, stream, k);
  return (int)hipGetLastError();
}

int ld_bf16_wgrad_launch(const WgradK& k, hipStream_t stream) {
  const int ntaps = k.KH * k.KW;
  if (ld_bf16_wgrad_tiled(k.Cout, k.Cin, k.Pout)) {
    const int blocks =
        ((k.Cout + 127) / 128) * ((k.Cin + 127) / 128) * ntaps * k.splits;
    LD_LAUNCH(conv_wgrad_tile_bf16_kernel, dim3(blocks), dim3(256), 0, stream,
                       k);
    return (int)hipGetLastError();
  }
  const int blocks = ((k.Cout + 63) / 64) * ((k.Cin + 63) / 64) * ntaps * k.splits;
  // 16-byte operand loads where four consecutive positions are contiguous
  const char* env = getenv("LD_CONV_BF16_WGRAD_VEC");
  const bool allow = !(env && env[0] == '0');
  const bool vy = allow && k.Pout % 4 == 0 && ((uintptr_t)k.dy % 16) == 0;
  const bool vx = vy && k.KH == 1 && k.KW == 1 && k.g.stride == 1 && k.g.pad == 0 &&
                  k.Pin == k.Pout && ((uintptr_t)k.x % 16) == 0;
  // measured (profiles/r02_layers_bf16_c8.csv vs the VY-only run): both
  // operands vectorised (+3-5 % on the 1x1 layers) pays, dY alone next to a
  // per-element X (3x3: two geometries per lane) is 10 % SLOWER -- the kernel is
  // bound by the latency of its one-step-ahead loads, not by instruction count
  const bool vy_only = vy && !vx && env && env[0] == 'y';  // test hook
  if (vx)
    LD_LAUNCH((conv_wgrad_wave_bf16_kernel<true, true>), dim3(blocks), dim3(64),
                       0, stream, k);
  else if (vy_only)
    LD_LAUNCH((conv_wgrad_wave_bf16_kernel<true, false>), dim3(blocks), dim3(64),
                       0, stream, k);
  else
    LD_LAUNCH((conv_wgrad_wave_bf16_kernel<false, false>), dim3(blocks),
                       dim3(64), 0, stream, k);
  return (int)hipGetLastError();
}

extern "C" int ld_conv_to_c8(const float* x, int N, int C, int P, void* out,
                             ld_stream_t stream) {
  if (!x || !out || N < 1 || C < 8 || C % 8 != 0 || P < 1) return LD_EINVAL;
  LD_LAUNCH(to_c8_kernel, dim3((P + 255) / 256, C / 8, N), dim3(256), 0,
                     (hipStream_t)stream, x, C / 8, P, (uintx4*)out);
  return (int)hipGetLastError();
}

extern "C" size_t ld_conv_bf16_weight_image_elems(int Cout, int Cin, int KH, int KW,
                                                 int backward) {
  if (Cout < 1 || Cin < 1 || KH < 1 || KW < 1) return 0;
  return backward ? (size_t)KH * KW * k8_blocks(Cout) * Cin * 8
                  : (size_t)KH * KW * k8_blocks(Cin) * Cout * 8;
}

extern "C" int ld_conv_bf16_weight_transform(const float* w, int Cout, int Cin, int KH,
                                             int KW, void* wt_fwd, void* wt_bwd,
                                             ld_stream_t stream) {
  if (!w || Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || (!wt_fwd && !wt_bwd))
    return LD_EINVAL;
  size_t total = 0;
  if (wt_fwd) total = ld_conv_bf16_weight_image_elems(Cout, Cin, KH, KW, 0);
  if (wt_bwd) total = max(total, ld_conv_bf16_weight_image_elems(Cout, Cin, KH, KW, 1));
  LD_LAUNCH(conv_weight_transform_bf16_kernel,
                     dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, Cout, Cin, KH * KW, (__bf16*)wt_fwd,
                     (__bf16*)wt_bwd);
  return (int)hipGetLastError();
}

extern "C" int ld_conv_bf16_weight_transform_batch(const ld_wt_job_t* jobs,
                                                   const int32_t* block_job, int nblocks,
                                                   ld_stream_t stream) {
  if (!jobs || !block_job || nblocks < 1) return LD_EINVAL;
  LD_LAUNCH(conv_weight_transform_bf16_batch_kernel, dim3(nblocks), dim3(256),
                     0, (hipStream_t)stream, jobs, block_job);
  return (int)hipGetLastError();
}
