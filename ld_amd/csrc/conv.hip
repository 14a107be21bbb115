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
//   fwd/dgrad    conv_stream_kernel: LDS-free.  Both MFMA operand layouts are
//                directly loadable from the layouts above, so every wavefront
//                streams its own operands through a register ring of raw
//                buffer loads and owns a (TM*32) x (TN*32) accumulator tile;
//                waves never meet at a barrier.  Tile shape / split-K picked
//                per layer geometry by a timing autotuner (tiles-per-SIMD
//                rounding decides, see launch_stream).  conv_igemm_kernel is
//                the older LDS double-buffered 64x64 tiling, kept for the 7x7
//                stem (flat (ci,kh,kw) reduction) and k % 8 != 0.
//   wgrad        D[co][ci] per tap = sum_j dY[co][j] * X[ci][pos(j,tap)],
//                split over j.  conv_wgrad_wave_kernel: one wavefront per
//                workgroup, 64x64 tile, PRIVATE LDS transpose buffer, no
//                barriers; partial slabs [split][tap][co][ci] are written
//                coalesced and summed in fixed order (deterministic, no float
//                atomics) by conv_wgrad_reduce.
#include <hip/hip_runtime.h>
#include <type_traits>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "conv_common.h"

namespace {

// ------------------------------------------------------------ forward/dgrad
// Tile shape is a template parameter: BM x BNT block tile, BKT k-slice.  Four
// wavefronts as 2 x 2, each owning a (BM/2) x (BNT/2) sub-tile = TM x TN MFMA
// 32x32 tiles.  Large tiles (128x128) for big spatial extents; small ones
// (64x64) keep >= 2 workgroups per CU on the 50x84 / 25x42 stages where a
// 128x128 tiling would leave half the 256 CUs idle.
//
// KG > 1 = intra-block split-K: the block has KG groups of 4 wavefronts, group g
// owns k-steps g, g+KG, ... with its own LDS double buffer; the partial
// accumulators are summed through LDS before the (single) epilogue.  Used when
// the layer yields too few 64x64 tiles to give every SIMD several waves (the
// 50x84 / 25x42 stages: PMC shows the MFMA pipe only ~45 % busy at 2
// waves/SIMD); the tile count, and with it the fused epilogue, is unchanged.
template <int BM, int BNT, int BKT, int MODE, int KG, int WVM, int WVN>
__global__ __launch_bounds__(64 * WVM * WVN * KG) void conv_igemm_kernel(ConvK a) {
  // WVM x WVN wavefronts per k-group (2 x 2, or 1 x 2 for the 32 x 64 tile that
  // doubles the workgroup count on the small-spatial stages)
  constexpr int NT = 64 * WVM * WVN;          // threads per k-group
  constexpr int WM = BM / WVM, WN = BNT / WVN;  // wave tile
  constexpr int TM = WM / 32, TN = WN / 32;  // MFMA tiles per wave
  constexpr int A_PER = BKT * BM / NT;   // A floats per thread per step
  constexpr int B_PER = BKT * BNT / NT;  // B floats per thread per step
  static_assert(A_PER >= 1 && B_PER >= 1, "tile too small");
  constexpr int NBUF = 2;
  constexpr int GROUP_LDS = NBUF * BKT * (BM + BNT);  // floats per k-group
  constexpr int NACC = TM * TN * 16;                  // accumulators per thread
  static_assert(KG == 1 || (KG - 1) * NACC * NT <= KG * GROUP_LDS,
                "split-K reduction does not fit the tile LDS");
  __shared__ float lds[KG * GROUP_LDS + 2 * BM];
  const int kg = __builtin_amdgcn_readfirstlane((int)threadIdx.x / NT);
  float* As = lds + kg * GROUP_LDS;      // [NBUF][BKT][BM]
  float* Bs = As + NBUF * BKT * BM;      // [NBUF][BKT][BNT]

  const int t = threadIdx.x % NT, lane = t & 63, wave = t >> 6;
  const int wm = wave / WVN, wn = wave % WVN;
  const int mtiles = (a.Cout + BM - 1) / BM;
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (tile % mtiles) * BM;
  const int n0 = (tile / mtiles) * BNT;

  // ---- this thread's B column (spatial position) -----------------------
  // Everything about the column is resolved ONCE into registers (level
  // geometry included): the k-loop's gather is then pure register arithmetic
  // (a per-step lookup of a.g.lv[level] is a dependent vector load + wait).
  const int jb = n0 + (t % BNT);
  const bool jvalid = jb < a.J;
  int bHin = 0, bWin = 0, boff = 0, bh0 = 0, bw0 = 0;
  if (jvalid) {
    const int n = jb / a.Pout, p = jb - n * a.Pout;
    int bl, bho, bwo;
    locate_out(a.g, p, bl, bho, bwo);
    bHin = a.g.lv[bl].Hin;
    bWin = a.g.lv[bl].Win;
    boff = n * a.Cin * a.Pin + a.g.lv[bl].off_in;  // element index of (n, 0, level)
    if (MODE == 1) {
      bh0 = bho + a.ch0;
      bw0 = bwo + a.cw0;
    } else {
      bh0 = bho * a.g.stride - a.g.pad;
      bw0 = bwo * a.g.stride - a.g.pad;
    }
  }
  // input offset of tap (kh, kw) for this column (MODE 1: of the class's tap
  // (i, j)), or false when it falls in the padding
  auto tap_off = [&](int kh, int kw, int& off) -> bool {
    const int hi = bh0 + kh, wi = bw0 + kw;
    if (hi < 0 || hi >= bHin || wi < 0 || wi >= bWin) return false;
    off = boff + hi * bWin + wi;
    return true;
  };
  // first k row this thread loads for B / A: wave-uniform (a wave never spans
  // two row groups), pinned to SGPRs so the row offsets go in `soffset`
  static_assert(BNT >= 64, "B row groups must be wave-uniform");
  const int bk0 = __builtin_amdgcn_readfirstlane((t / BNT) * B_PER);
  // ---- this thread's A column (output channel) ------------------------
  const int am = t % BM;
  const bool avalid = (m0 + am) < a.Cout;
  // A row group = wave-uniform part (soffset) + lane part (only when BM < 64:
  // the two half-waves own different row groups; goes into voffset -- safe on
  // the A side because the weight image is zero-padded to 32 k-rows)
  const int ak0s = __builtin_amdgcn_readfirstlane(((wave * 64) / BM) * A_PER);
  const int ak0v = (BM < 64) ? (lane / BM) * A_PER : 0;
  const int ak0 = ak0s + ak0v;

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // wave-uniform scalars pinned to SGPRs
  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int KW = __builtin_amdgcn_readfirstlane(a.KW);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int ntaps = __builtin_amdgcn_readfirstlane(
      MODE == 1 ? a.nth * a.ntw : a.KH * a.KW);
  const int csteps = (Cin + BKT - 1) / BKT;
  const int ktot = Cin * ntaps;  // MODE 2: flat (ci, kh, kw) reduction index
  const int nsteps = (MODE == 2) ? (ktot + BKT - 1) / BKT : ntaps * csteps;
  float a_st[A_PER], b_st[B_PER];

  // Tile loads are raw buffer loads: one 32-bit per-lane voffset (kOOB when the
  // element is padding / outside the tile -> the hardware returns 0), the row
  // (k) offsets wave-uniform in soffset.  No branches, no selects on loaded
  // values, 2 address VGPRs per tile side: nothing forces a vmcnt wait before
  // the MFMAs, so the loads of step s+1 fly under the MFMAs of step s.
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t rw = make_rsrc(a.wt, a.wt_bytes);
  const int Kpad = __builtin_amdgcn_readfirstlane(a.Kpad);
  const unsigned va =
      avalid ? (unsigned)(m0 + am + ak0v * Cout) * 4u : kOOB;
  auto load_tile = [&](int step, float* ra, float* rb) {
    if (MODE == 2) {
      // small-Cin (stem) im2col: every k row has its own (ci, kh, kw)
      const int k0 = step * BKT;
#pragma unroll
      for (int i = 0; i < A_PER; ++i)
        ra[i] = buf_load(rw, va, (unsigned)(k0 + ak0s + i) * Cout * 4u);
#pragma unroll
      for (int i = 0; i < B_PER; ++i) {
        const int k = k0 + bk0 + i;          // scalar
        const int ci = min(k / ntaps, Cin - 1);
        const int r = k % ntaps;
        const int kh = r / KW, kw = r - kh * KW;
        int off = 0;
        const bool kb = jvalid && tap_off(kh, kw, off);
        rb[i] = buf_load(rx, kb ? (unsigned)off * 4u : kOOB,
                           (unsigned)ci * Pin * 4u);
      }
      return;
    }
    int tap = step / csteps;
    const int ci0 = (step - tap * csteps) * BKT;
    int kh, kw;
    if (MODE == 1) {  // class-local tap (i, j) -> weight-image tap
      const int ntw = __builtin_amdgcn_readfirstlane(a.ntw);
      kh = tap / ntw;
      kw = tap - kh * ntw;
      tap = (a.kh0 + 2 * kh) * KW + a.kw0 + 2 * kw;
    } else {
      kh = tap / KW;
      kw = tap - kh * KW;
    }
    int off = 0;
    const bool ok = jvalid && tap_off(kh, kw, off);
    const unsigned vb = ok ? (unsigned)off * 4u : kOOB;
    const unsigned sa = (unsigned)(tap * Kpad + ci0 + ak0s) * Cout * 4u;
#pragma unroll
    for (int i = 0; i < A_PER; ++i)
      ra[i] = buf_load(rw, va, sa + (unsigned)i * Cout * 4u);
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
      // k-tail rows re-read the last valid channel (finite); their A rows are
      // the zero padding of the weight image
      const int row = min(ci0 + bk0 + i, Cin - 1);
      rb[i] = buf_load(rx, vb, (unsigned)row * Pin * 4u);
    }
  };
  auto store_tile = [&](int buf, const float* ra, const float* rb) {
    float* ap = As + buf * BKT * BM + ak0 * BM + am;
#pragma unroll
    for (int i = 0; i < A_PER; ++i) ap[i * BM] = ra[i];
    float* bp = Bs + buf * BKT * BNT + bk0 * BNT + (t % BNT);
#pragma unroll
    for (int i = 0; i < B_PER; ++i) bp[i * BNT] = rb[i];
  };

  const int l31 = lane & 31, lk = lane >> 5;
  constexpr int KP = BKT / 2;
  // one k-step of MFMAs on LDS buffer `buf`, fragment reads one k-pair ahead
  auto compute = [&](int buf) {
    const float* ap = As + buf * BKT * BM + wm * WM + l31;
    const float* bp = Bs + buf * BKT * BNT + wn * WN + l31;
    float af[2][TM], bf[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[0][i] = ap[lk * BM + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[0][j] = bp[lk * BNT + j * 32];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const int c = kp & 1;
      if (kp + 1 < KP) {
        const int kr = 2 * (kp + 1) + lk;
#pragma unroll
        for (int i = 0; i < TM; ++i) af[c ^ 1][i] = ap[kr * BM + i * 32];
#pragma unroll
        for (int j = 0; j < TN; ++j) bf[c ^ 1][j] = bp[kr * BNT + j * 32];
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i], bf[c][j],
                                                           acc[i][j], 0, 0, 0);
    }
    // pin the interleave (0x100 = DS read, 0x008 = MFMA; LLVM SchedGroupMask)
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      if (kp + 1 < KP) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, TM * TN, 0);
    }
  };

  // ---- software-pipelined main loop (KG == 1) ---------------------------
  // Per k-step the plain loop pays, between the last MFMA of tile s and the
  // first of tile s+1: ds_write + barrier + first ds_read latency -- a bubble
  // that only other resident workgroups can fill (PMC: pipe 45 % busy at ~2
  // waves/SIMD).  Here instead
  //   * global loads run two tiles ahead (two register stages),
  //   * tile s+1 is written to LDS in the MIDDLE of tile s's MFMAs,
  //   * the barrier and the first fragment reads of tile s+1 are issued before
  //     the last two k-pairs of tile s, whose MFMAs cover that latency.
  constexpr bool PIPE = (KG == 1) && (MODE != 2) && (KP >= 8);
  bool use_pipe = PIPE && a.pipe;
  if (use_pipe) {
    float a_s2[A_PER], b_s2[B_PER];
    float f0a[TM], f0b[TN];  // fragments of k-pair 0 of the current tile
    constexpr int WPOS = KP / 2;  // k-pair after which tile s+1 goes to LDS
    auto frag0 = [&](int buf) {
      const float* ap = As + buf * BKT * BM + wm * WM + l31;
      const float* bp = Bs + buf * BKT * BNT + wn * WN + l31;
#pragma unroll
      for (int i = 0; i < TM; ++i) f0a[i] = ap[lk * BM + i * 32];
#pragma unroll
      for (int j = 0; j < TN; ++j) f0b[j] = bp[lk * BNT + j * 32];
    };
    // one pipelined k-step: tile in `buf`; `wa/wb` = register stage holding
    // tile s+1 (written to buf^1 mid-step when has_next)
    auto pstep = [&](int buf, bool has_next, const float* wa, const float* wb) {
      const float* ap = As + buf * BKT * BM + wm * WM + l31;
      const float* bp = Bs + buf * BKT * BNT + wn * WN + l31;
      float af[2][TM], bf[2][TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) af[0][i] = f0a[i];
#pragma unroll
      for (int j = 0; j < TN; ++j) bf[0][j] = f0b[j];
#pragma unroll
      for (int kp = 0; kp < KP; ++kp) {
        const int c = kp & 1;
        if (kp + 1 < KP) {
          const int kr = 2 * (kp + 1) + lk;
#pragma unroll
          for (int i = 0; i < TM; ++i) af[c ^ 1][i] = ap[kr * BM + i * 32];
#pragma unroll
          for (int j = 0; j < TN; ++j) bf[c ^ 1][j] = bp[kr * BNT + j * 32];
        }
        if (kp == WPOS && has_next) store_tile(buf ^ 1, wa, wb);
        if (kp == KP - 2) {
          // every fragment of this tile has been requested; after the barrier
          // nobody reads `buf` any more and tile s+1 is complete in buf^1
          __syncthreads();
          if (has_next) frag0(buf ^ 1);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i], bf[c][j],
                                                             acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    load_tile(0, a_st, b_st);
    store_tile(0, a_st, b_st);
    if (nsteps > 1) load_tile(1, a_s2, b_s2);  // stage B <- tile 1
    __syncthreads();
    frag0(0);
    for (int step = 0; step < nsteps; step += 2) {
      // even step: tile `step` in buffer 0; tile step+1 waits in stage B
      if (step + 2 < nsteps) load_tile(step + 2, a_st, b_st);
      pstep(0, step + 1 < nsteps, a_s2, b_s2);
      if (step + 1 < nsteps) {
        // odd step: tile step+1 in buffer 1; tile step+2 waits in stage A
        if (step + 3 < nsteps) load_tile(step + 3, a_s2, b_s2);
        pstep(1, step + 2 < nsteps, a_st, b_st);
      }
    }
  } else {
  // group kg runs k-steps kg, kg+KG, ...; every group executes the same number
  // of barriers
  const int iters = (nsteps + KG - 1) / KG;
  if (kg < nsteps) {
    load_tile(kg, a_st, b_st);
    store_tile(0, a_st, b_st);
  }
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    const int cur = it & 1;
    const int step = kg + it * KG;
    const bool more = step + KG < nsteps;
    if (more) load_tile(step + KG, a_st, b_st);
    if (step < nsteps) compute(cur);
    if (more) store_tile(cur ^ 1, a_st, b_st);
    __syncthreads();
  }
  }
  __syncthreads();

  // ---- epilogue: C/D layout col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
  // Per-channel affine / bias are staged through LDS once per block (the
  // k-loop's LDS is free now), so the store loop has no dependent global
  // loads; without an affine the defaults (1, 0) make it branch-free.
  float* s_scale = lds + KG * GROUP_LDS;  // [BM]
  float* s_shift = s_scale + BM;          // [BM]  shift (+ bias)
  if (KG > 1 && kg > 0) {
    // partial accumulators -> LDS, [group-1][reg][thread] (conflict-free)
    float* red = lds + (size_t)(kg - 1) * NACC * NT + t;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          red[((i * TN + j) * 16 + r) * NT] = acc[i][j][r];
  }
  if (kg == 0 && t < BM) {
    const int co = m0 + t;
    float sc = 1.0f, sh = 0.0f;
    if (co < a.Cout) {
      if (a.scale) {
        sc = a.scale[co];
        sh = a.shift[co];
      }
      if (a.bias) sh += a.bias[co];
    }
    s_scale[t] = sc;
    s_shift[t] = sh;
  }
  __syncthreads();
  if (KG > 1) {
    if (kg > 0) return;
#pragma unroll
    for (int g = 1; g < KG; ++g) {
      const float* red = lds + (size_t)(g - 1) * NACC * NT + t;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            acc[i][j][r] += red[((i * TN + j) * 16 + r) * NT];
    }
  }
  const bool has_res = a.residual != nullptr;
  const bool relu = a.relu != 0;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int jc = n0 + wn * WN + j * 32 + l31;
    if (jc >= a.J) continue;
    const int n = jc / a.Pout;
    int p = jc - n * a.Pout;
    int prow = a.Pout;  // positions per (n, c) row of y
    if (MODE == 1) {    // compact class position -> full output position
      int l, hc, wc;
      locate_out(a.g, p, l, hc, wc);
      p = a.foff[l] + (2 * hc + a.ph) * a.fW[l] + 2 * wc + a.pw;
      prow = a.Pfull;
    }
    const size_t colbase = (size_t)n * a.Cout * prow + p;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int rbase = wm * WM + i * 32 + 4 * lk;
      float res[16];
      if (has_res) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = rbase + (r & 3) + 8 * (r >> 2);
          res[r] = (m0 + row < a.Cout)
                       ? a.residual[colbase + (size_t)(m0 + row) * prow]
                       : 0.0f;
        }
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (m0 + row >= a.Cout) continue;
        float v = acc[i][j][r] * s_scale[row] + s_shift[row];
        if (MODE == 0 && a.y_raw)
          a.y_raw[colbase + (size_t)(m0 + row) * prow] = acc[i][j][r];
        if (has_res) v += res[r];
        if (relu) v = fmaxf(v, 0.0f);
        a.y[colbase + (size_t)(m0 + row) * prow] = v;
      }
    }
  }
}

// ------------------------------------------------- forward/dgrad, streaming
// LDS-free variant.  The 32x32x2 MFMA operand layouts are directly loadable:
//   A (weights) lane l <- Wt[tap][ci + (l>>5)][co0 + (l&31)]   2 x 128 B rows
//   B (pixels)  lane l <- X[n][ci + (l>>5)][pos(j0 + (l&31))]  2 x 128 B rows
// so LDS is not needed as a layout transformer, and fp32 MFMA is slow enough
// (256 flop/clk/CU) that a wavefront owning a (TM*32) x (TN*32) register tile
// needs only (TM+TN)*256 B per TM*TN*64 MFMA cycles from L1/L2 -- 16 B/clk/CU
// at 2x2, a quarter of the L1 rate.  Every wavefront therefore streams its
// own operands through a D-deep register ring (vmcnt-tracked, D k-pairs in
// flight) and never meets another wavefront at a barrier: the MFMA pipe of a
// SIMD stays busy as long as ANY resident wave has operands, instead of the
// whole workgroup stalling on ds_write -> s_barrier -> ds_read every k-step.
// The four waves of a workgroup sit on neighbouring tiles only so that their
// shared A / B rows hit in the CU's L1.
//
// KS = 4: the four waves of a workgroup share ONE wave tile and split its
// reduction (chunk c goes to wave c % 4); the partial accumulators meet once, in
// LDS, before the epilogue.  Layers whose tile count is only ~1-5 per SIMD
// (J = 8400 / 2100 stages) lose up to half the machine to the rounding of
// tiles-per-SIMD; quartering the work unit brings that back (8.2 -> 9 instead
// of 2.05 -> 3).  The per-element summation order differs from KS = 1.
// DBG (timing attribution only, LD_STREAM_DBG, results wrong when non-zero): 1 = the
// ring is never refilled after the prologue, 2 = pixel (B) loads read nothing,
// 4 = weight (A) loads read nothing, 8 = no epilogue stores / epilogue loads.
template <int TM, int TN, int WVM, int MODE, int D, int OCC, int KS, int DBG = 0>
__global__ __launch_bounds__(256, OCC) void conv_stream_kernel(ConvK a) {
  static_assert(KS == 1 || (KS == 4 && WVM == 1), "KS is 1 or 4");
  static_assert(D * (TM + TN) < 64, "ring exceeds the 6-bit vmcnt counter");
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
  const int Kpad = __builtin_amdgcn_readfirstlane(a.Kpad);
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
      boff[j] = n * Cin * Pin + a.g.lv[bl].off_in + lk * Pin;
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
    va[i] = co < Cout ? (unsigned)(lk * Cout + co) * 4u : kOOB;
  }
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t rw = make_rsrc(a.wt, a.wt_bytes);

  // load cursor: (tap, ci) of the next chunk of D k-pairs
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
  float ra[D][TM], rb[D][TN];
  auto load_kp = [&](int d, unsigned sa, unsigned sb) {
#pragma unroll
    for (int i = 0; i < TM; ++i) ra[d][i] = buf_load(rw, (DBG & 4) ? kOOB : va[i], sa);
#pragma unroll
    for (int j = 0; j < TN; ++j) rb[d][j] = buf_load(rx, (DBG & 2) ? kOOB : vb[j], sb);
  };

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  auto mfma_kp = [&](int d) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[d][i], rb[d][j],
                                                         acc[i][j], 0, 0, 0);
  };

  const int cchunks = Cin / (2 * D);  // host guarantees Cin % (2*D) == 0
  const int nchunks = ntaps * cchunks;
  // this wave's chunks: kslice, kslice + KS, ...
  const int mychunks = nchunks > kslice ? (nchunks - kslice + KS - 1) / KS : 0;
  int ltap = 0, lci = 0;
  auto advance = [&](int n) {
    lci += 2 * D * n;
    const int t0 = ltap;
    while (lci >= Cin) {
      lci -= Cin;
      ++ltap;
    }
    if (ltap != t0 && ltap < ntaps) set_tap(ltap);
  };
  set_tap(0);
  if (KS > 1) advance(kslice);
  if (mychunks > 0) {
    {
      const unsigned sa = (unsigned)(wtap * Kpad + lci) * Cout * 4u;
      const unsigned sb = (unsigned)lci * Pin * 4u;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        // same issue order as the steady state, so the loop-header vmcnt merge
        // stays exact (4 * (D - 1) outstanding)
        load_kp(d, sa + (unsigned)(2 * d) * Cout * 4u,
                sb + (unsigned)(2 * d) * Pin * 4u);
        __builtin_amdgcn_sched_barrier(0);
      }
      advance(KS);
    }
    for (int c = 0; c + 1 < mychunks; ++c) {
      const unsigned sa = (unsigned)(wtap * Kpad + lci) * Cout * 4u;
      const unsigned sb = (unsigned)lci * Pin * 4u;
#pragma unroll
      for (int d = 0; d < D; ++d) {
        // pin the ring order: without the fences the scheduler sinks all D
        // refills below the MFMAs and the prefetch distance collapses to zero
        mfma_kp(d);
        __builtin_amdgcn_sched_barrier(0);
        if (!(DBG & 1))
          load_kp(d, sa + (unsigned)(2 * d) * Cout * 4u,
                  sb + (unsigned)(2 * d) * Pin * 4u);
        __builtin_amdgcn_sched_barrier(0);
      }
      advance(KS);
    }
#pragma unroll
    for (int d = 0; d < D; ++d) mfma_kp(d);
  }
  if (KS == 4) {
    // pairwise tree through LDS, fixed order ((w0 + w2) + (w1 + w3)); two
    // slots instead of three keep the 2x2 tile at 32 KB per workgroup
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
  // Round 3, after reading the ISA of the loop this replaces (r-major, plain
  // pointers, `if (flag)` around every optional operand): the pointers of ConvK
  // may alias as far as hipcc knows, so every affine / bias / residual load
  // stayed behind the previous store and was waited for with vmcnt(0) -- ~2
  // dependent round trips per output element, more cycles than the whole k-loop
  // of a 256-deep 1x1 conv.  Now:
  //   * buffer descriptors for every operand; an absent one has extent 0 (its
  //     loads return 0, no memory traffic), rows >= Cout and columns >= J use
  //     the out-of-range offset (stores dropped): no branches, no exec masks;
  //   * a row group = 4 accumulator rows x all TN column tiles: its 12 + 4 TN
  //     loads are in flight together, then its stores -- one round trip per
  //     group (TM * 4 per tile) and fewer registers than the 32 affine values
  //     the old loop held (1 x 1 tile: 76 -> 48 VGPRs, 6 -> 8 waves per SIMD);
  //   * the residual / raw-output variants are separate straight-line copies.
  // Same arithmetic per element as before: bit-identical results.
  if (DBG & 8) {  // keep the accumulators alive without the epilogue's traffic
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1.2345e-30f) a.y[0] = s;
    return;
  }
  const bool has_res = a.residual != nullptr;
  const bool has_raw = MODE == 0 && a.y_raw != nullptr;
  const bool relu = a.relu != 0;
  const bool has_aff = a.scale != nullptr, has_bias = a.bias != nullptr;
  const int prow = __builtin_amdgcn_readfirstlane(MODE == 1 ? a.Pfull : a.Pout);
  const unsigned ybytes = (unsigned)a.N * (unsigned)Cout * (unsigned)prow * 4u;  // host: < 2 GiB
  const rsrc_t r_y = make_rsrc(a.y, ybytes);
  const rsrc_t r_raw = make_rsrc(a.y_raw, has_raw ? ybytes : 0u);
  const rsrc_t r_res = make_rsrc(a.residual, has_res ? ybytes : 0u);
  const rsrc_t r_sc = make_rsrc(a.scale, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_sh = make_rsrc(a.shift, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_bi = make_rsrc(a.bias, has_bias ? (unsigned)Cout * 4u : 0u);
  // per column tile: byte offset of (row m0 + 4 lk, this lane's column)
  unsigned voff[TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int jc = n0 + j * 32 + l31;
    const int n = jc / a.Pout;
    int p = jc - n * a.Pout;
    if (MODE == 1 && jc < a.J) {
      int l, hc, wc;
      locate_out(a.g, p, l, hc, wc);
      p = a.foff[l] + (2 * hc + a.ph) * a.fW[l] + 2 * wc + a.pw;
    }
    voff[j] = jc < a.J ? ((unsigned)(n * Cout + m0 + 4 * lk) * (unsigned)prow + (unsigned)p) * 4u
                       : kOOB;
  }
  const unsigned prow4 = (unsigned)prow * 4u;
  auto run = [&](auto res_c, auto raw_c) {
    constexpr bool RES = decltype(res_c)::value, RAW = decltype(raw_c)::value;
#pragma unroll
    for (int ig = 0; ig < TM * 4; ++ig) {
      const int i = ig >> 2, g = ig & 3;
      const int rl = i * 32 + 8 * g;  // row of the group - (m0 + 4 lk)
      // scheduling fence: without it hipcc front-loads the loads of every group
      // (nothing orders them any more) and the epilogue, not the k-loop, sets
      // the kernel's register count
      __builtin_amdgcn_sched_barrier(0);
      float sc[4], sh[4], bi[4], rv[TN][4];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned ro = (unsigned)(m0 + 4 * lk + rl + e) * 4u;  // >= Cout: zeros
        sc[e] = buf_load(r_sc, ro, 0);
        sh[e] = buf_load(r_sh, ro, 0);
        bi[e] = buf_load(r_bi, ro, 0);
      }
      if (RES) {
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = m0 + 4 * lk + rl + e < Cout;
            rv[j][e] = buf_load(r_res, ok ? voff[j] : kOOB, (unsigned)(rl + e) * prow4);
          }
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = m0 + 4 * lk + rl + e < Cout;
        const float scale = has_aff ? sc[e] : 1.0f;
        const float shift = sh[e] + bi[e];  // absent operands loaded as 0
        const unsigned so = (unsigned)(rl + e) * prow4;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int r = 4 * g + e;
          const unsigned vo = ok ? voff[j] : kOOB;
          // (a copy: __builtin_bit_cast applied to the vector ELEMENT expression
          // acc[i][j][r] reads element 0 with this hipcc -- seen in the ISA)
          const float av = acc[i][j][r];
          float v = av * scale + shift;
          if (RAW)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, av), r_raw,
                                                  vo, so, 0);
          if (RES) v += rv[j][e];
          v = relu ? fmaxf(v, 0.0f) : v;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r_y, vo,
                                                so, 0);
        }
      }
    }
  };
  if (has_res) {
    if (has_raw) run(std::true_type{}, std::true_type{});
    else run(std::true_type{}, std::false_type{});
  } else {
    if (has_raw) run(std::false_type{}, std::true_type{});
    else run(std::false_type{}, std::false_type{});
  }
}

// ------------------------------------ 1x1 convs: vector operand loads -------
// PMC of the R101 teacher's 256 -> 1024 1x1 conv at 50x84 (round 3,
// profiles/r03_pmc_teacher_1x1_256_1024.txt: 2.39 GHz, MFMA busy 43 %, TA busy
// 59 % of the kernel): with 32x32 wave tiles the streaming kernel issues two
// 4-byte wave loads per MFMA and each costs the texture-address unit ~9 cycles
// -- 73 TA cycles per 64 MFMA cycles per CU: the VECTOR-MEMORY ISSUE RATE, not
// the matrix pipe, not HBM, bounds the small-tile shapes the tuner has to pick
// on the 50x84 / 25x42 stages.  For a 1x1, stride-1 conv the B operand has no
// taps: the GEMM column j IS the input position, so one lane can take VEC
// consecutive positions of a channel row in ONE 8/16-byte load and feed VEC
// MFMA column tiles from it -- tile e of a wave = the columns {j0 + VEC*l + e}
// (a strided set; the GEMM does not care, and the epilogue then owns VEC
// consecutive positions per accumulator row: 8/16-byte stores and residual
// loads, cdna_hip_programming.md T21).  Loads per k-pair and wave: TM + 1 for
// TM*VEC MFMAs (1x1 tile before: 2 per MFMA).  Same reduction order (channel
// ascending, one accumulator per output): same bits as the KS = 1 shapes.
// Table code: ks = 2, tn = VEC.
template <int VEC>
struct VecF;
template <>
struct VecF<2> { typedef float T __attribute__((ext_vector_type(2))); };
template <>
struct VecF<4> { typedef float T __attribute__((ext_vector_type(4))); };

template <int VEC>
__device__ __forceinline__ typename VecF<VEC>::T buf_load_vec(rsrc_t r, unsigned voff,
                                                             unsigned soff) {
  typedef typename VecF<VEC>::T V;
  if constexpr (VEC == 2)
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
  else
    return __builtin_bit_cast(V, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

template <int TM, int VEC, int WVM, int D>
__global__ __launch_bounds__(256, 2) void conv1x1_vec_kernel(ConvK a) {
  static_assert(D * (TM + 1) < 64, "ring exceeds the 6-bit vmcnt counter");
  typedef typename VecF<VEC>::T V;
  constexpr int WVN = 4 / WVM;
  constexpr int WM = TM * 32, WN = VEC * 32;
  constexpr int BM = WVM * WM, BNT = WVN * WN;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave / WVN, wn = wave % WVN;
  const int l31 = lane & 31, lk = lane >> 5;
  const int mtiles = (a.Cout + BM - 1) / BM;
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int m0 = (tile % mtiles) * BM + wm * WM;
  const int n0 = (tile / mtiles) * BNT + wn * WN;
  if (m0 >= a.Cout || n0 >= a.J) return;  // waves are independent: no barriers

  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int P = __builtin_amdgcn_readfirstlane(a.Pin);  // == Pout

  // this lane's VEC consecutive columns (P % VEC == 0: never across images)
  const int jg = n0 + VEC * l31;
  const bool jok = jg < a.J;
  const int n = jok ? jg / P : 0, p = jok ? jg - (jg / P) * P : 0;
  const unsigned vb = jok ? (unsigned)(n * Cin * P + p + lk * P) * 4u : kOOB;
  unsigned va[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int co = m0 + i * 32 + l31;
    va[i] = co < Cout ? (unsigned)(lk * Cout + co) * 4u : kOOB;
  }
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t rw = make_rsrc(a.wt, a.wt_bytes);

  float ra[D][TM];
  V rb[D];
  auto load_kp = [&](int d, unsigned sa, unsigned sb) {
#pragma unroll
    for (int i = 0; i < TM; ++i) ra[d][i] = buf_load(rw, va[i], sa);
    rb[d] = buf_load_vec<VEC>(rx, vb, sb);
  };
  floatx16 acc[TM][VEC];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int e = 0; e < VEC; ++e)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][e][r] = 0.0f;
  auto mfma_kp = [&](int d) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        acc[i][e] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[d][i], rb[d][e], acc[i][e],
                                                         0, 0, 0);
  };
  const int nchunks = Cin / (2 * D);  // host guarantees Cin % (2*D) == 0
#pragma unroll
  for (int d = 0; d < D; ++d) {
    load_kp(d, (unsigned)(2 * d) * Cout * 4u, (unsigned)(2 * d) * P * 4u);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int c = 1; c < nchunks; ++c) {
    const unsigned sa = (unsigned)(c * 2 * D) * Cout * 4u;
    const unsigned sb = (unsigned)(c * 2 * D) * P * 4u;
#pragma unroll
    for (int d = 0; d < D; ++d) {
      mfma_kp(d);
      __builtin_amdgcn_sched_barrier(0);
      load_kp(d, sa + (unsigned)(2 * d) * Cout * 4u, sb + (unsigned)(2 * d) * P * 4u);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) mfma_kp(d);

  // ---- epilogue: VEC consecutive positions per accumulator row ---------------
  if (!jok) return;
  const bool has_res = a.residual != nullptr;
  const bool relu = a.relu != 0;
  const bool has_aff = a.scale != nullptr, has_bias = a.bias != nullptr;
  const size_t colbase = (size_t)n * Cout * P + p;
  const rsrc_t r_sc = make_rsrc(a.scale, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_sh = make_rsrc(a.shift, has_aff ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_bi = make_rsrc(a.bias, has_bias ? (unsigned)Cout * 4u : 0u);
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rbase = m0 + i * 32 + 4 * lk;
    // groups of RG rows, every load of a group before its first store (see the
    // streaming kernel's epilogue); RG x VEC residual registers
    constexpr int RG = 2;
#pragma unroll
    for (int g = 0; g < 16 / RG; ++g) {
      float sc[RG], sh[RG], bi[RG];
      V q[RG];
      // group fence: this group's loads stay behind the previous group's stores
      // and none is consumed before all are issued (absent affine / bias: extent-0
      // descriptors, loaded as 0)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < RG; ++e) {
        const int r = RG * g + e;
        const int row = min(rbase + (r & 3) + 8 * (r >> 2), Cout - 1);
        sc[e] = buf_load(r_sc, (unsigned)row * 4u, 0);
        sh[e] = buf_load(r_sh, (unsigned)row * 4u, 0);
        bi[e] = buf_load(r_bi, (unsigned)row * 4u, 0);
        if (has_res)
          q[e] = *reinterpret_cast<const V*>(a.residual + colbase + (size_t)row * P);
      }
#pragma unroll
      for (int e = 0; e < RG; ++e) {
        const int r = RG * g + e;
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row >= Cout) continue;
        const size_t o = colbase + (size_t)row * P;
        V raw, v;
#pragma unroll
        for (int c = 0; c < VEC; ++c) {
          raw[c] = acc[i][c][r];
          v[c] = raw[c] * (has_aff ? sc[e] : 1.0f) + (sh[e] + bi[e]);
        }
        if (a.y_raw) *reinterpret_cast<V*>(a.y_raw + o) = raw;
        if (has_res) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) v[c] += q[e][c];
        }
        if (relu) {
#pragma unroll
          for (int c = 0; c < VEC; ++c) v[c] = fmaxf(v[c], 0.0f);
        }
        *reinterpret_cast<V*>(a.y + o) = v;
      }
    }
  }
}

// ---------------------------------------------- the 7x7 stem, streaming ----
// Small-Cin forward conv (the ResNet stem: 3 -> 64, 7x7, stride 2, pad 3 --
// resnet.py:536-548) with the FLAT reduction index k = (ci, kh, kw), in the
// same LDS-free design as conv_stream_kernel: round 2 ran it on the older LDS
// tiling (conv_igemm_kernel MODE 2) at 49.6 TFLOP/s, 0.2 ms per launch.
//   A (weights) lane l <- Wt[k + (l>>5)][co0 + (l&31)]     image [Kpad][Cout],
//               zero rows for k >= Cin*KH*KW (so the k-tail needs no masking)
//   B (pixels)  lane l <- X[n][ci][ho*S - P + kh][wo*S - P + kw] for ITS k =
//               2*step + (l>>5): each half-wave walks its own (ci, kh, kw)
//               counter in steps of 2 -- a few VALU ops per k-pair against
//               TM*TN*64 MFMA cycles
// Four waves of a workgroup own four neighbouring column tiles (all of Cout
// each: TM = Cout / 32), D k-pairs of operands in flight per wave.
template <int TM, int TN, int D>
__global__ __launch_bounds__(256, 2) void conv_stem_kernel(ConvK a) {
  static_assert(D * (TM + TN) < 64, "ring exceeds the 6-bit vmcnt counter");
  constexpr int WN = TN * 32;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int l31 = lane & 31, lk = lane >> 5;
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int n0 = (tile * 4 + wave) * WN;
  if (n0 >= a.J) return;  // waves are independent: no barriers below
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int KH = __builtin_amdgcn_readfirstlane(a.KH);
  const int KW = __builtin_amdgcn_readfirstlane(a.KW);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int K = __builtin_amdgcn_readfirstlane(a.Cin * a.KH * a.KW);
  const int npairs = __builtin_amdgcn_readfirstlane(a.Kpad / 2);

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
      boff[j] = n * a.Cin * Pin + a.g.lv[bl].off_in;
      bh0[j] = bho * a.g.stride - a.g.pad;
      bw0[j] = bwo * a.g.stride - a.g.pad;
    }
  }
  unsigned va[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int co = i * 32 + l31;
    va[i] = co < Cout ? (unsigned)(lk * Cout + co) * 4u : kOOB;
  }
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t rw = make_rsrc(a.wt, a.wt_bytes);

  // this half-wave's reduction cursor: k = 2 * step + lk -> (ci, kh, kw)
  int kk = lk, kci = 0, kkh = 0, kkw = lk;
  if (kkw >= KW) {  // KW == 1
    kkw -= KW;
    if (++kkh >= KH) {
      kkh = 0;
      ++kci;
    }
  }
  auto step2 = [&]() {  // advance the cursor by two
    kk += 2;
    kkw += 2;
    while (kkw >= KW) {
      kkw -= KW;
      if (++kkh >= KH) {
        kkh = 0;
        ++kci;
      }
    }
  };
  float ra[D][TM], rb[D][TN];
  auto load_kp = [&](int d, int pair) {
    const unsigned sa = (unsigned)(2 * pair) * (unsigned)Cout * 4u;
#pragma unroll
    for (int i = 0; i < TM; ++i) ra[d][i] = buf_load(rw, va[i], sa);
    const bool kin = kk < K;
    const int cbase = kci * Pin;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int hi = bh0[j] + kkh, wi = bw0[j] + kkw;
      const bool ok = kin && hi >= 0 && hi < bHin[j] && wi >= 0 && wi < bWin[j];
      const unsigned vb = ok ? (unsigned)(boff[j] + cbase + hi * bWin[j] + wi) * 4u : kOOB;
      rb[d][j] = buf_load(rx, vb, 0);
    }
    step2();
  };
  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  auto mfma_kp = [&](int d) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[d][i], rb[d][j],
                                                         acc[i][j], 0, 0, 0);
  };
  // npairs is a multiple of D (Kpad is a multiple of 32, D <= 16)
  const int nchunks = npairs / D;
#pragma unroll
  for (int d = 0; d < D; ++d) {
    load_kp(d, d);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (int c = 1; c < nchunks; ++c) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      mfma_kp(d);
      __builtin_amdgcn_sched_barrier(0);
      load_kp(d, c * D + d);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) mfma_kp(d);

  // ---- epilogue (as conv_stream_kernel's, no residual on the stem) ----------
  const bool relu = a.relu != 0;
  const bool has_aff = a.scale != nullptr, has_bias = a.bias != nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rbase = i * 32 + 4 * lk;
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = min(rbase + (r & 3) + 8 * (r >> 2), Cout - 1);
      sc[r] = has_aff ? a.scale[row] : 1.0f;
      sh[r] = has_aff ? a.shift[row] : 0.0f;
      if (has_bias) sh[r] += a.bias[row];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int jc = n0 + j * 32 + l31;
      if (jc >= a.J) continue;
      const int n = jc / a.Pout;
      const int p = jc - n * a.Pout;
      const size_t colbase = (size_t)n * Cout * a.Pout + p;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        if (row >= Cout) continue;
        float v = acc[i][j][r] * sc[r] + sh[r];
        if (a.residual) v += a.residual[colbase + (size_t)row * a.Pout];
        if (relu) v = fmaxf(v, 0.0f);
        a.y[colbase + (size_t)row * a.Pout] = v;
      }
    }
  }
}

// ------------------------------------- the 7x7 stem, operands through LDS ----
// Round 6.  conv_stem_kernel above gathers every B element with its own 4-byte
// load: per k-pair a wave issues TM + TN = 4 gather instructions for 4 MFMAs, 8
// waves per CU keep the CU's address unit busy ~2x longer than the matrix pipes
// (60-62 TFLOP/s in every round's layer table).  Here a workgroup owns TH x 64
// output positions of ONE image (one output row segment per wave), stages the
// input patch ((TH - 1) S + KH rows x 63 S + KW columns x Cin) and the whole
// [Kpad][Cout] weight image in LDS with coalesced loads, and the k-loop reads
// both operands with ds_read_b32 (A: 32 consecutive channels per half-wave; B:
// stride-S positions, 2-way bank conflicts at S = 2, hidden under the 64-cycle
// MFMAs).  The reduction runs over the same k pairs in the same order with the
// same operand values as conv_stem_kernel: results are bit-identical
// (tests/test_gpu_layers.py).  One level only (the stem has one).
// (First version, measured: 233 us per launch against the gather kernel's 164 --
// its staging loop walked the patch with one dependent load per iteration.  The
// geometry is a template now: every staging load of a thread is in flight at once,
// the k-loop is fully unrolled with compile-time LDS offsets.)
template <int TH, int CIN, int KH, int KW, int S>
__global__ __launch_bounds__(TH * 64) void conv_stem_lds_kernel(ConvK a, int tiles_w,
                                                               int tiles_h) {
  constexpr int TM = 2, TN = 2;  // 64 channels x 64 positions per wave
  constexpr int NT = TH * 64;    // one wave per output row of the tile
  constexpr int COUT = 64;
  constexpr int K = CIN * KH * KW;
  constexpr int KPAD = (K + kKPad - 1) / kKPad * kKPad;
  constexpr int PR = (TH - 1) * S + KH;  // patch rows
  constexpr int PC = 63 * S + KW;        // patch columns
  constexpr int PCP = PC + 1;            // row pitch in LDS
  constexpr int TOT = CIN * PR * PC;
  constexpr int NIT = (TOT + NT - 1) / NT;
  constexpr int NW4 = KPAD * COUT / 4 / NT;  // float4 weight loads per thread
  static_assert(KPAD * COUT % (4 * NT) == 0, "weight image in whole 16-byte rounds");
  __shared__ __attribute__((aligned(16))) float wsm[KPAD * COUT];  // [Kpad][Cout]
  __shared__ float psm[CIN * PR * PCP];                            // [Cin][PR][PCP]
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, lk = lane >> 5;
  const int pad = __builtin_amdgcn_readfirstlane(a.g.pad);
  const int Hin = __builtin_amdgcn_readfirstlane(a.g.lv[0].Hin);
  const int Win = __builtin_amdgcn_readfirstlane(a.g.lv[0].Win);
  const int Hout = __builtin_amdgcn_readfirstlane(a.g.lv[0].Hout);
  const int Wout = __builtin_amdgcn_readfirstlane(a.g.lv[0].Wout);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tw = b % tiles_w;
  b /= tiles_w;
  const int th = b % tiles_h;
  const int n = b / tiles_h;
  const int ho0 = th * TH, wo0 = tw * 64;
  // ---- stage: every load of this thread issued before the first LDS write
  {
    const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
    const rsrc_t rw = make_rsrc(a.wt, a.wt_bytes);
    typedef unsigned uintx4_ __attribute__((ext_vector_type(4)));
    uintx4_ wv[NW4];
#pragma unroll
    for (int i = 0; i < NW4; ++i)
      wv[i] = __builtin_bit_cast(
          uintx4_, __builtin_amdgcn_raw_buffer_load_b128(rw, (unsigned)(t + i * NT) * 16u, 0, 0));
    const int h00 = ho0 * S - pad, w00 = wo0 * S - pad;
    const int xbase = n * CIN * Pin + a.g.lv[0].off_in;
    float pv[NIT];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = t + i * NT;
      const int c = e % PC, r = (e / PC) % PR, ci = e / (PC * PR);
      const int hi = h00 + r, wi = w00 + c;
      const bool ok = (e < TOT) & ((unsigned)hi < (unsigned)Hin) & ((unsigned)wi < (unsigned)Win);
      pv[i] = buf_load(rx, ok ? (unsigned)(xbase + ci * Pin + hi * Win + wi) * 4u : kOOB, 0);
    }
#pragma unroll
    for (int i = 0; i < NW4; ++i) ((uintx4_*)wsm)[t + i * NT] = wv[i];
#pragma unroll
    for (int i = 0; i < NIT; ++i) {
      const int e = t + i * NT;
      const int c = e % PC, r = (e / PC) % PR, ci = e / (PC * PR);
      if (e < TOT) psm[(ci * PR + r) * PCP + c] = pv[i];
    }
  }
  __syncthreads();
  const int ho = ho0 + wave;  // this wave's output row
  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  const float* wl = wsm + lk * COUT + l31;
  const float* pl = psm + (wave * S) * PCP + l31 * S;
#pragma unroll
  for (int pr = 0; pr < KPAD / 2; ++pr) {
    // k = 2 pr + lk -> (ci, kh, kw): compile-time for either half-wave
    constexpr auto off_of = [](int k) {
      const int ci = k / (KH * KW), kh = (k / KW) % KH, kw = k % KW;
      return (ci * PR + kh) * PCP + kw;
    };
    const int k0 = 2 * pr, k1 = 2 * pr + 1;
    float ra[TM], rb[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) ra[i] = wl[(2 * pr) * COUT + i * 32];
    const int o0 = k0 < K ? off_of(k0) : 0, o1 = k1 < K ? off_of(k1) : 0;
    const bool kin = lk ? (k1 < K) : (k0 < K);
    const float* pk = pl + (lk ? o1 : o0);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const float v = pk[j * 32 * S];
      rb[j] = kin ? v : 0.0f;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[i], rb[j], acc[i][j], 0, 0, 0);
  }
  if (ho >= Hout) return;
  // ---- epilogue (conv_stem_kernel's arithmetic)
  const bool relu = a.relu != 0;
  const bool has_aff = a.scale != nullptr, has_bias = a.bias != nullptr;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int rbase = i * 32 + 4 * lk;
    float sc[16], sh[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = rbase + (r & 3) + 8 * (r >> 2);
      sc[r] = has_aff ? a.scale[row] : 1.0f;
      sh[r] = has_aff ? a.shift[row] : 0.0f;
      if (has_bias) sh[r] += a.bias[row];
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int wo = wo0 + j * 32 + l31;
      if (wo >= Wout) continue;
      const int p = a.g.lv[0].off_out + ho * Wout + wo;
      const size_t colbase = (size_t)n * COUT * a.Pout + p;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        float v = acc[i][j][r] * sc[r] + sh[r];
        if (a.residual) v += a.residual[colbase + (size_t)row * a.Pout];
        if (relu) v = fmaxf(v, 0.0f);
        a.y[colbase + (size_t)row * a.Pout] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ wgrad --
__global__ __launch_bounds__(kThreads, 2) void conv_wgrad_kernel(WgradK a) {
  constexpr int BM = 128, BNc = 128;  // co x ci tile
  constexpr int ROWS_PER = BM / 8;    // rows per thread (8 row-groups of 32 lanes)
  __shared__ float lds[2 * (BM + BNc) * WLD];
  __shared__ int s_geo[LD_MAX_LEVELS * 6];  // level table (LDS: cheap per-step lookups)
  float* As = lds;                       // [2][BM][WLD]   dY rows (co), k = j
  float* Bs = lds + 2 * BM * WLD;        // [2][BNc][WLD]  X rows (ci), k = j

  const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Pout = __builtin_amdgcn_readfirstlane(a.Pout);
  const int nlev = __builtin_amdgcn_readfirstlane(a.g.num_levels);
  const int stride = __builtin_amdgcn_readfirstlane(a.g.stride);
  const int pad = __builtin_amdgcn_readfirstlane(a.g.pad);
  const int mt = (Cout + BM - 1) / BM, nt = (Cin + BNc - 1) / BNc;
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

  const int kq = t & (WBK - 1);  // this thread's k (column j offset) in a step
  const int r0 = t >> 5;         // 0..7: first row; rows r0 + 8*i

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t ry = make_rsrc(a.dy, a.dy_bytes);
  float a_st[ROWS_PER], b_st[ROWS_PER];

  // buffer loads: per-lane voffset (kOOB -> 0), row offsets 8*i*P in soffset
  auto load_tile = [&](int j0) {
    const int j = j0 + kq;
    const bool jok = j < jend;
    unsigned vy = kOOB, vx = kOOB;
    if (jok) {
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
#pragma unroll
    for (int i = 0; i < ROWS_PER; ++i) {
      a_st[i] = buf_load(ry, (m0 + r0 + 8 * i < Cout) ? vy : kOOB,
                         (unsigned)(8 * i) * Pout * 4u);
      b_st[i] = buf_load(rx, (c0 + r0 + 8 * i < Cin) ? vx : kOOB,
                         (unsigned)(8 * i) * Pin * 4u);
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
  constexpr int KP = WBK / 2;
  for (int step = 0; step < nsteps; ++step) {
    const int cur = step & 1;
    if (step + 1 < nsteps) load_tile(jbeg + (step + 1) * WBK);
    const float* ap = As + cur * BM * WLD + (wm * 64 + l31) * WLD;
    const float* bp = Bs + cur * BNc * WLD + (wn * 64 + l31) * WLD;
    float af[2][2], bf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[0][i] = ap[i * 32 * WLD + lk];
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[0][j] = bp[j * 32 * WLD + lk];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const int c = kp & 1;
      if (kp + 1 < KP) {
        const int kc = 2 * (kp + 1) + lk;
#pragma unroll
        for (int i = 0; i < 2; ++i) af[c ^ 1][i] = ap[i * 32 * WLD + kc];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[c ^ 1][j] = bp[j * 32 * WLD + kc];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i], bf[c][j],
                                                           acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      if (kp + 1 < KP) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    if (step + 1 < nsteps) store_tile(cur ^ 1);
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

// ---------------------------------------------------- wgrad, wave-private --
// Same decoupling as the streaming forward kernel, for the GEMM whose operands
// cannot be loaded in MFMA layout (both dY and X are contiguous along the
// reduction index j, the MFMA wants a lane per ROW): every wavefront is its own
// workgroup, owns a 64(co) x 64(ci) tile of one tap and one j-split, and
// transposes its operand tiles through a PRIVATE LDS buffer -- global -> regs
// (coalesced along j) -> LDS [row][j] -> fragment reads [row = lane][k].  LDS
// traffic of one wave is ordered by the hardware, so no s_barrier exists
// anywhere: a SIMD's MFMA pipe idles only when all of its resident waves are
// between tiles at the same moment.  The next tile's global loads are in flight
// under the 64 MFMAs of the current one.
template <int BKJ>
__global__ __launch_bounds__(64, BKJ == 32 ? 2 : 3) void conv_wgrad_wave_kernel(WgradK a) {
  constexpr int TB = 64;         // tile edge (co and ci)
  constexpr int LDW = BKJ + 1;   // odd row stride: conflict-free fragment reads
  constexpr int RG = 64 / BKJ;   // rows covered by one load instruction
  constexpr int RPER = TB / RG;  // rows per lane and operand
  __shared__ float lds[2 * TB * LDW];
  __shared__ int s_geo[LD_MAX_LEVELS * 6];
  float* As = lds;             // [TB][LDW]  dY rows (co)
  float* Bs = lds + TB * LDW;  // [TB][LDW]  X rows (ci)

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

  const int kq = lane % BKJ;  // this lane's j offset within a step
  const int r0 = lane / BKJ;  // first row; rows r0 + RG*i

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t ry = make_rsrc(a.dy, a.dy_bytes);
  float a_st[RPER], b_st[RPER];
  const bool rows_full = (m0 + TB <= Cout) && (c0 + TB <= Cin);

  auto load_tile = [&](int j0) {
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
      vy = (unsigned)(n * Cout * Pout + (m0 + r0) * Pout + p) * 4u;
      if (hi >= 0 && hi < Hin && wi >= 0 && wi < Win)
        vx = (unsigned)(n * Cin * Pin + (c0 + r0) * Pin + s_geo[l * 6 + 4] +
                        hi * Win + wi) * 4u;
    }
    // row offsets advance in SGPRs inside the step: 2 * RPER loop-invariant
    // soffsets would be hoisted, exhaust the SGPR file, get parked in VGPRs and
    // come back as waterfall loops around every load (seen in the ISA)
    unsigned da = (unsigned)RG * Pout * 4u, db = (unsigned)RG * Pin * 4u;
    asm volatile("" : "+s"(da), "+s"(db));
    unsigned sa = 0, sb = 0;
    if (rows_full) {
#pragma unroll
      for (int i = 0; i < RPER; ++i) {
        a_st[i] = buf_load(ry, vy, sa);
        b_st[i] = buf_load(rx, vx, sb);
        sa += da;
        sb += db;
      }
    } else {
      // rows past the channel count: per-lane row budgets compared inside the
      // step (hoisted lane masks would again spill the SGPR file)
      int na = (Cout - m0 - r0 + RG - 1) / RG, nb = (Cin - c0 - r0 + RG - 1) / RG;
      asm volatile("" : "+v"(na), "+v"(nb));
#pragma unroll
      for (int i = 0; i < RPER; ++i) {
        a_st[i] = buf_load(ry, i < na ? vy : kOOB, sa);
        b_st[i] = buf_load(rx, i < nb ? vx : kOOB, sb);
        sa += da;
        sb += db;
      }
    }
  };
  auto store_tile = [&]() {
    float* ap = As + r0 * LDW + kq;
    float* bp = Bs + r0 * LDW + kq;
#pragma unroll
    for (int i = 0; i < RPER; ++i) {
      ap[RG * i * LDW] = a_st[i];
      bp[RG * i * LDW] = b_st[i];
    }
  };

  const int nsteps = (jend - jbeg + BKJ - 1) / BKJ;
  const int l31 = lane & 31, lk = lane >> 5;
  constexpr int KP = BKJ / 2;
  if (nsteps > 0) load_tile(jbeg);
  for (int step = 0; step < nsteps; ++step) {
    store_tile();
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (step + 1 < nsteps) load_tile(jbeg + (step + 1) * BKJ);
    const float* ap = As + l31 * LDW;
    const float* bp = Bs + l31 * LDW;
    float af[2][2], bf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) af[0][i] = ap[i * 32 * LDW + lk];
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[0][j] = bp[j * 32 * LDW + lk];
#pragma unroll
    for (int kp = 0; kp < KP; ++kp) {
      const int c = kp & 1;
      if (kp + 1 < KP) {
        const int kc = 2 * (kp + 1) + lk;
#pragma unroll
        for (int i = 0; i < 2; ++i) af[c ^ 1][i] = ap[i * 32 * LDW + kc];
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[c ^ 1][j] = bp[j * 32 * LDW + kc];
      }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c][i], bf[c][j],
                                                           acc[i][j], 0, 0, 0);
    }
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
  // eight slab reads in flight per thread, summed in split order
  float s = 0.0f;
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(k + u) * per + i];
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < splits; ++k) s += slabs[(size_t)k * per + i];
  const size_t o = ((size_t)co * Cin + ci) * ntaps + tap;
  dw[o] = accumulate ? dw[o] + s : s;
}

// Same sum, four consecutive ci per thread (Cin % 4 == 0): 16-byte slab loads,
// eight in flight, each component added in split order -- the same bits as the
// scalar kernel.  (rocprofv3 of the fp32 step: 60 reduce launches, 2.45 ms per
// step at ~0.8 TB/s with the scalar loop.)
__global__ __launch_bounds__(256) void conv_wgrad_reduce4_kernel(
    const float* __restrict__ slabs, int splits, int ntaps, int Cout, int Cin,
    float* __restrict__ dw, int accumulate) {
  const size_t per = (size_t)ntaps * Cout * Cin;
  const size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i >= per) return;
  const int ci = (int)(i % Cin);
  const size_t q = i / Cin;
  const int co = (int)(q % Cout), tap = (int)(q / Cout);
  floatx4 s = {0.0f, 0.0f, 0.0f, 0.0f};
  int k = 0;
  for (; k + 8 <= splits; k += 8) {
    floatx4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      v[u] = *reinterpret_cast<const floatx4*>(slabs + (size_t)(k + u) * per + i);
#pragma unroll
    for (int u = 0; u < 8; ++u) s += v[u];
  }
  for (; k < splits; ++k)
    s += *reinterpret_cast<const floatx4*>(slabs + (size_t)k * per + i);
  if (ntaps == 1) {
    float* o = dw + (size_t)co * Cin + ci;
    if (accumulate) s += *reinterpret_cast<const floatx4*>(o);
    *reinterpret_cast<floatx4*>(o) = s;
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const size_t o = ((size_t)co * Cin + ci + e) * ntaps + tap;
      dw[o] = accumulate ? dw[o] + s[e] : s[e];
    }
  }
}

// (Cout, Cin, KH, KW) -> fwd image [tap][Cin_pad][Cout] and dgrad image
// [KH*KW-1-tap][Cout_pad][Cin]; the pad rows (k >= Cin resp. Cout) are zero.
__device__ __forceinline__ void weight_transform_at(const float* __restrict__ w,
                                                    int Cout, int Cin, int ntaps,
                                                    int CinPad, int CoutPad,
                                                    float* __restrict__ wt_fwd,
                                                    float* __restrict__ wt_bwd,
                                                    size_t i) {
  if (wt_fwd && i < (size_t)ntaps * CinPad * Cout) {
    // i enumerates the fwd image [tap][ci][co] (coalesced writes)
    const int co = (int)(i % Cout);
    const size_t q = i / Cout;
    const int ci = (int)(q % CinPad), tap = (int)(q / CinPad);
    wt_fwd[i] = ci < Cin ? w[((size_t)co * Cin + ci) * ntaps + tap] : 0.0f;
  }
  if (wt_bwd && i < (size_t)ntaps * CoutPad * Cin) {
    // i enumerates the bwd image [tapf][co][ci]
    const int ci = (int)(i % Cin);
    const size_t q = i / Cin;
    const int co = (int)(q % CoutPad), tapf = (int)(q / CoutPad);
    wt_bwd[i] = co < Cout
                    ? w[((size_t)co * Cin + ci) * ntaps + (ntaps - 1 - tapf)]
                    : 0.0f;
  }
}

__global__ void conv_weight_transform_kernel(const float* __restrict__ w, int Cout,
                                             int Cin, int ntaps, int CinPad,
                                             int CoutPad,
                                             float* __restrict__ wt_fwd,
                                             float* __restrict__ wt_bwd) {
  weight_transform_at(w, Cout, Cin, ntaps, CinPad, CoutPad, wt_fwd, wt_bwd,
                      (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ __launch_bounds__(256) void conv_weight_transform_batch_kernel(
    const ld_wt_job_t* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  const ld_wt_job_t j = jobs[block_job[blockIdx.x]];
  const int CinPad = (j.Cin + kKPad - 1) / kKPad * kKPad;
  const int CoutPad = (j.Cout + kKPad - 1) / kKPad * kKPad;
  weight_transform_at(j.w, j.Cout, j.Cin, j.ntaps, CinPad, CoutPad, j.wt_fwd, j.wt_bwd,
                      (size_t)(blockIdx.x - j.first_block) * 256 + threadIdx.x);
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

// Tile selection (measured on MI355X, profiles/r01_kernels_s4.json): the 64x64
// tile wins or ties on every layer shape of the LD step -- finer granularity
// over the 256 CUs beats the larger tile's operand reuse at fp32-MFMA rates --
// with a 32-deep k-slice once the reduction is long enough to amortise it.
// LD_CONV_TILE=BMxBNxBK overrides (benchmarking).
struct TileCfg { int bm, bn, bk; };

inline int tile_blocks(const ConvK& k, const TileCfg& c) {
  return ((k.Cout + c.bm - 1) / c.bm) * ((k.J + c.bn - 1) / c.bn);
}

inline TileCfg pick_tile(const ConvK& k) {
  if (const char* env = getenv("LD_CONV_TILE")) {
    TileCfg c{0, 0, 0};
    if (sscanf(env, "%dx%dx%d", &c.bm, &c.bn, &c.bk) == 3) return c;
  }
  const int ktot = k.Cin * k.KH * k.KW;
  return TileCfg{64, 64, ktot >= 512 ? 32 : 16};
}

inline int set_extents(ConvK& k, size_t x_floats, size_t wt_floats) {
  // the output (and the residual / raw output of its shape) is addressed through
  // buffer descriptors too (round 3)
  const size_t y_floats = (size_t)k.N * k.Cout * (size_t)max(k.Pout, k.Pfull);
  if (x_floats * 4 >= (size_t)kOOB || wt_floats * 4 >= (size_t)kOOB ||
      y_floats * 4 >= (size_t)kOOB)
    return LD_EUNSUPPORTED;  // 32-bit buffer offsets: tensors must be < 2 GiB
  k.x_bytes = (unsigned)(x_floats * 4);
  k.wt_bytes = (unsigned)(wt_floats * 4);
  return 0;
}

// ---- streaming kernel dispatch + per-shape autotune -----------------------
// Shapes are "TM x TN x WVM x D" (wave tile in 32s, waves along Cout, ring
// depth).  Which one wins is decided by how the tile count divides over the
// 1024 SIMDs (profiles/r01_kernels_s20_stream.json: 2x2 reaches 0.83 of peak
// on the head towers but 0.44 on the 50x84 stages where 1x1 reaches 0.57), so
// the first launch of every distinct layer geometry times the candidates on
// the caller's buffers (the launch is idempotent) and caches the winner -- the
// same contract as cudnn.benchmark=True, which mmdet sets for these configs
// (mmdet/apis/train.py).  All candidates accumulate each output element in the
// same order (tap-major, channel ascending, one accumulator), so the choice
// never changes a single bit of the result.
//   LD_CONV_STREAM = "0"           LDS kernel only
//   LD_CONV_STREAM = "2x2x2x8"     force a shape
//   LD_CONV_AUTOTUNE = "0"         model-based pick, no timing
//   LD_CONV_TUNE_LOG = "1"         print the picks to stderr
struct StreamCfg {
  int tm, tn, wvm, d, ks;
};
#define LD_STREAM_SHAPES(X)                                                        \
  X(2, 2, 2, 8, 1) X(2, 2, 1, 8, 1) X(2, 1, 2, 8, 1) X(1, 2, 2, 8, 1)              \
  X(1, 2, 4, 8, 1) X(1, 2, 1, 8, 1) X(1, 1, 2, 8, 1) X(1, 1, 1, 8, 1)              \
  X(1, 1, 4, 8, 1) X(3, 1, 1, 8, 1) X(3, 2, 1, 8, 1) X(2, 2, 2, 4, 1)              \
  X(2, 2, 1, 4, 1) X(1, 2, 2, 4, 1) X(1, 1, 2, 4, 1) X(1, 1, 1, 4, 1)              \
  X(3, 1, 1, 4, 1) X(3, 2, 1, 4, 1) X(1, 1, 1, 8, 4) X(1, 2, 1, 8, 4)              \
  X(2, 1, 1, 8, 4) X(2, 2, 1, 8, 4)                                                \
  X(1, 1, 1, 16, 1) X(1, 1, 2, 16, 1) X(1, 1, 4, 16, 1) X(2, 1, 2, 16, 1)          \
  X(1, 2, 2, 16, 1) X(2, 1, 1, 16, 1) X(1, 1, 1, 16, 4) X(2, 1, 1, 16, 4)          \
  X(1, 2, 1, 16, 4)
// ks = 2: conv1x1_vec_kernel (1x1 stride-1 convs only), tn = positions per lane
#define LD_VEC_SHAPES(X)                                                           \
  X(1, 4, 1, 8, 2) X(1, 4, 2, 8, 2) X(1, 4, 4, 8, 2) X(2, 4, 1, 8, 2)              \
  X(2, 4, 2, 8, 2) X(1, 2, 1, 8, 2) X(1, 2, 2, 8, 2) X(1, 2, 4, 8, 2)              \
  X(2, 2, 2, 8, 2) X(1, 4, 2, 16, 2) X(1, 4, 4, 16, 2) X(1, 2, 4, 16, 2)
constexpr StreamCfg kStreamCfgs[] = {
#define LD_STREAM_ROW(TM_, TN_, WVM_, D_, KS_) {TM_, TN_, WVM_, D_, KS_},
    LD_STREAM_SHAPES(LD_STREAM_ROW) LD_VEC_SHAPES(LD_STREAM_ROW)
#undef LD_STREAM_ROW
};
constexpr int kNumStreamCfgs = sizeof(kStreamCfgs) / sizeof(kStreamCfgs[0]);

// Residency cap (round 3).  A 1x1 / 1x2 wave tile needs 72-92 VGPRs, so 5-7
// workgroups fit a CU and a layer of ~8 tiles per SIMD (the R101 teacher's 50x84
// stages) runs as ONE synchronous round: every wave is in its prologue at the
// same time, then all compete for the matrix pipe, then all reach the epilogue
// together -- 68 MB of residual reads + stores with no wave left in its k-loop
// (VERDICT round 2 weak #5).  Holding the launch to `cap`
// workgroups per CU (by reserving dynamic LDS it never touches) makes the
// dispatcher hand out the tiles in several rounds that drift apart, so one
// workgroup's epilogue runs under its neighbours' MFMAs.  cap is a tuned field
// of the shape table; it never changes a bit of the result.
inline unsigned cap_lds_bytes(int cap, int static_bytes) {
  // per-workgroup LDS so that exactly `cap` workgroups fit the CU's 160 KiB
  // (<= 64 KiB: no opt-in attribute needed)
  static const int kb[] = {0, 64, 64, 48, 36, 28, 24};
  if (cap < 2 || cap > 6) return 0;
  const int want = kb[cap] * 1024 - static_bytes;
  return want > 0 ? (unsigned)want : 0;
}

template <int MODE>
int launch_stream_cfg(const ConvK& k, const StreamCfg& c, hipStream_t stream,
                      int cap = 0) {
  if (const char* env = getenv("LD_ALLOW_WRONG_RESULTS") ? getenv("LD_STREAM_DBG") : nullptr) {
    // timing attribution (tools/stream_dbg.py; the variants compute WRONG results,
    // hence the second variable): the 1x1x1 d16 ks4 shape only
    const int dbg = atoi(env);
    if (MODE == 0 && dbg && c.tm == 1 && c.tn == 1 && c.wvm == 1 && c.d == 16) {
      const int bm = 32, bn = 32;
      const int nb = ((k.Cout + bm - 1) / bm) * ((k.J + bn - 1) / bn);
#define LD_DBG_CASE(V_)                                                                  \
  if (dbg == V_ && c.ks == 4) {                                                          \
    LD_LAUNCH((conv_stream_kernel<1, 1, 1, 0, 16, 4, 4, V_>), dim3(nb), dim3(256), \
                       0, stream, k);                                                    \
    return (int)hipGetLastError();                                                       \
  }
      LD_DBG_CASE(1) LD_DBG_CASE(2) LD_DBG_CASE(4) LD_DBG_CASE(6) LD_DBG_CASE(8) LD_DBG_CASE(9)
      LD_DBG_CASE(14)
#undef LD_DBG_CASE
    }
  }
  const int bm = (c.ks == 4 ? 1 : c.wvm) * c.tm * 32;
  const int bn = (c.ks == 4 ? 1 : 4 / c.wvm) * c.tn * 32;
  const int nb = ((k.Cout + bm - 1) / bm) * ((k.J + bn - 1) / bn);
  const unsigned lds =
      cap_lds_bytes(cap, c.ks == 4 ? 2 * c.tm * c.tn * 16 * 64 * 4 : 4);
#define LD_STREAM_CASE(TM_, TN_, WVM_, D_, KS_)                                    \
  if (c.tm == TM_ && c.tn == TN_ && c.wvm == WVM_ && c.d == D_ && c.ks == KS_) {   \
    LD_LAUNCH((conv_stream_kernel<TM_, TN_, WVM_, MODE, D_, (KS_ == 4 ? 4 : 2), KS_>), \
                       dim3(nb), dim3(256), lds, stream, k);                       \
    return (int)hipGetLastError();                                                 \
  }
  if (c.ks != 2) {
    LD_STREAM_SHAPES(LD_STREAM_CASE)
  }
#undef LD_STREAM_CASE
  if constexpr (MODE == 0) {
#define LD_VEC_CASE(TM_, TN_, WVM_, D_, KS_)                                       \
  if (c.ks == 2 && c.tm == TM_ && c.tn == TN_ && c.wvm == WVM_ && c.d == D_) {     \
    LD_LAUNCH((conv1x1_vec_kernel<TM_, TN_, WVM_, D_>), dim3(nb), dim3(256), lds, \
                       stream, k);                                                 \
    return (int)hipGetLastError();                                                 \
  }
    LD_VEC_SHAPES(LD_VEC_CASE)
#undef LD_VEC_CASE
  }
  return LD_EUNSUPPORTED;
}

#define MODE1_TAPS(k) ((k).nth > 0 ? (k).nth * (k).ntw : (k).KH * (k).KW)
inline bool stream_cfg_fits(const ConvK& k, const StreamCfg& c) {
  if (c.wvm < 1) return false;
  if (k.Cin % (2 * c.d) != 0) return false;
  if (c.ks == 2) {
    // vector 1x1 kernel: no taps, no stride, columns == input positions, and
    // every row of x / y / residual / y_raw aligned for tn-float accesses
    if (k.KH != 1 || k.KW != 1 || k.g.stride != 1 || k.g.pad != 0 || k.nth > 0 ||
        k.Pin != k.Pout || k.Pout % c.tn != 0 || k.x_c8 || k.y_c8 || k.res_c8)
      return false;
    const uintptr_t m = (uintptr_t)(c.tn * 4 - 1);
    if ((((uintptr_t)k.x | (uintptr_t)k.y | (uintptr_t)k.residual |
          (uintptr_t)k.y_raw) & m) != 0)
      return false;
    const int bmv = c.wvm * c.tm * 32, cout32v = (k.Cout + 31) / 32 * 32;
    if (c.wvm > 1 && bmv > cout32v) return false;
    if (c.wvm == 1 && c.tm * 32 >= cout32v + 32) return false;
    return true;
  }
  if (c.d == 4 && k.Cin % 16 == 0) return false;  // the 8-deep ring covers it
  const int bm = c.wvm * c.tm * 32;
  const int cout32 = (k.Cout + 31) / 32 * 32;
  if (c.ks == 4) {
    // split-K only pays when the reduction is long enough to quarter
    const int nchunks = (MODE1_TAPS(k)) * (k.Cin / (2 * c.d));
    if (nchunks < 16) return false;
  }
  if (c.wvm > 1 && bm > cout32) return false;  // whole waves of padding rows
  if (c.wvm == 1 && c.tm * 32 >= cout32 + 32) return false;
  return true;
}

// model-based pick: workgroup rounds over 256 CUs x per-shape pipe efficiency
inline int stream_cfg_model(const ConvK& k) {
  int best = -1;
  double best_t = 0;
  for (int i = 0; i < kNumStreamCfgs; ++i) {
    const StreamCfg& c = kStreamCfgs[i];
    if (c.ks == 2) continue;  // the vector 1x1 shapes are used where TUNED
    if (!stream_cfg_fits(k, c)) continue;
    const int bm = (c.ks == 4 ? 1 : c.wvm) * c.tm * 32;
    const int bn = (c.ks == 4 ? 1 : 4 / c.wvm) * c.tn * 32;
    const long nb = (long)((k.Cout + bm - 1) / bm) * ((k.J + bn - 1) / bn);
    const int area = c.tm * c.tn;
    const double eff = (area >= 4 ? 0.90 : area >= 2 ? 0.84 : 0.79) *
                       (c.ks == 4 ? 0.97 : 1.0);
    const int occ = area >= 4 ? 4 : 5;  // resident workgroups per CU
    const double rounds =
        nb <= 256L * occ ? (double)((nb + 255) / 256) : (double)nb / 256.0;
    // work per workgroup: 4 wave tiles, or one when its K is split four ways
    const double t = rounds * (c.ks == 4 ? 1 : 4) * area / eff;
    if (best < 0 || t < best_t) {
      best = i;
      best_t = t;
    }
  }
  return best;
}

constexpr int kTuneReps = 3;

inline int stream_cfg_index(const LdTuneCfg& c) {
  for (int i = 0; i < kNumStreamCfgs; ++i)
    if (kStreamCfgs[i].tm == c.tm && kStreamCfgs[i].tn == c.tn &&
        kStreamCfgs[i].wvm == c.wvm && kStreamCfgs[i].d == c.d &&
        kStreamCfgs[i].ks == c.ks)
      return i;
  return -1;
}

// Shape choice of a launch: forced by LD_CONV_STREAM, else the tuning table
// (ld_conv_tune_load / ld_conv_tune_*), else the round-count model -- a pure
// function of the geometry.  Never times anything and never synchronises: the
// launch entry points only enqueue.  *forced gets the LD_CONV_STREAM override.
template <int MODE>
int pick_stream_cfg(const ConvK& k, StreamCfg* forced, int* cap) {
  *cap = 0;
  if (k.Cin % 8 != 0) return -1;
  if (const char* env = getenv("LD_CONV_STREAM")) {
    if (env[0] == '0' && env[1] == 0) return -1;
    StreamCfg c;
    c.ks = 1;
    if (sscanf(env, "%dx%dx%dx%dx%dx%d", &c.tm, &c.tn, &c.wvm, &c.d, &c.ks, cap) >= 4) {
      while (c.d > 4 && k.Cin % (2 * c.d) != 0) c.d /= 2;
      *forced = c;
      return -2;
    }
  }
  LdTuneCfg t;
  if (ld_tune_lookup(make_tune_key(MODE, 0, k), &t)) {
    const int i = stream_cfg_index(t);
    if (i >= 0 && stream_cfg_fits(k, kStreamCfgs[i])) {
      *cap = t.cap;
      return i;
    }
  }
  return stream_cfg_model(k);
}

template <int MODE>
int launch_stream(const ConvK& k, hipStream_t stream) {
  StreamCfg forced;
  int cap = 0;
  const int pick = pick_stream_cfg<MODE>(k, &forced, &cap);
  if (pick == -1) return LD_EUNSUPPORTED;
  if (pick == -2 && forced.ks == 2 && !stream_cfg_fits(k, forced)) {
    // a forced vector-1x1 shape on a conv it cannot serve (taps, stride,
    // alignment): the model's pick instead
    const int m = stream_cfg_model(k);
    return m < 0 ? LD_EUNSUPPORTED : launch_stream_cfg<MODE>(k, kStreamCfgs[m], stream);
  }
  if (pick == -2) {
    const int rc = launch_stream_cfg<MODE>(k, forced, stream, cap);
    if (rc != LD_EUNSUPPORTED) return rc;
    forced.ks = 1;  // no split-K instance at this ring depth
    const int rc1 = launch_stream_cfg<MODE>(k, forced, stream, cap);
    if (rc1 != LD_EUNSUPPORTED) return rc1;
    const int m = stream_cfg_model(k);  // no such instance for this layer
    return m < 0 ? LD_EUNSUPPORTED : launch_stream_cfg<MODE>(k, kStreamCfgs[m], stream);
  }
  return launch_stream_cfg<MODE>(k, kStreamCfgs[pick], stream, cap);
}

// Explicit tuning (ld_conv_tune_forward / ld_conv_tune_dgrad): times every
// candidate shape of the geometry on the caller's buffers (the launch is
// idempotent), records the winner in the table.  This is the ONLY place that
// synchronises; returns 0 (tuned), 1 (already in the table / nothing to tune).
template <int MODE>
int tune_stream(const ConvK& k, hipStream_t stream) {
  if (k.Cin % 8 != 0) return 1;
  const LdTuneKey key = make_tune_key(MODE, 0, k);
  LdTuneCfg have;
  if (ld_tune_lookup(key, &have)) return 1;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cap);
  if (cap != hipStreamCaptureStatusNone) return LD_EUNSUPPORTED;
  int pick = stream_cfg_model(k);
  if (pick < 0) return 1;
  // quiesce the device first: work queued on other streams (the teacher's
  // forward) would otherwise share the CUs with some candidates and not others
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best_ms = -1.0f;
  int best_cap = 0;
  // residency caps worth timing: none, and 2-4 workgroups per CU.  A layer
  // with more than ~16 workgroups per CU is in steady state anyway.
  static const int kCaps[] = {0, 4, 3, 2};
  for (int i = 0; i < kNumStreamCfgs; ++i) {
    if (!stream_cfg_fits(k, kStreamCfgs[i])) continue;
    for (int ci = 0; ci < 4; ++ci) {
      const int cap_try = kCaps[ci];
      if (launch_stream_cfg<MODE>(k, kStreamCfgs[i], stream, cap_try) != 0) break;
      float ms = -1.0f;
      for (int trial = 0; trial < 2; ++trial) {  // best of two: clocks wander
        (void)hipEventRecord(e0, stream);
        for (int rep = 0; rep < kTuneReps; ++rep)
          launch_stream_cfg<MODE>(k, kStreamCfgs[i], stream, cap_try);
        (void)hipEventRecord(e1, stream);
        if (hipEventSynchronize(e1) != hipSuccess) break;
        float t = 0.0f;
        (void)hipEventElapsedTime(&t, e0, e1);
        if (ms < 0.0f || t < ms) ms = t;
      }
      if (ms < 0.0f) continue;
      // a cap must win by > 1 %: ties go to the plain launch
      if (best_ms < 0.0f || ms < best_ms * (cap_try ? 0.99f : 1.0f)) {
        best_ms = ms;
        pick = i;
        best_cap = cap_try;
      }
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  const StreamCfg& c = kStreamCfgs[pick];
  if (const char* lg = getenv("LD_CONV_TUNE_LOG"))
    if (lg[0] == '1') {
      const double fl = 2.0 * k.J * k.Cout * k.Cin *
                        (MODE == 1 ? k.nth * k.ntw : k.KH * k.KW);
      fprintf(stderr,
              "[ld_conv] mode %d Cin %d Cout %d k %dx%d s%d J %d lv %d -> "
              "%dx%dx%dx%dx%d cap %d  %.1f TFLOP/s\n",
              MODE, k.Cin, k.Cout, k.KH, k.KW, k.g.stride, k.J, k.g.num_levels,
              c.tm, c.tn, c.wvm, c.d, c.ks, best_cap,
              best_ms > 0 ? fl / (best_ms * 1e-3 / kTuneReps) / 1e12 : 0.0);
    }
  if (best_ms > 0.0f)
    ld_tune_store(key, LdTuneCfg{c.tm, c.tn, c.wvm, c.d, c.ks, best_cap});
  return 0;
}

template <int MODE>
int launch_igemm(const ConvK& k_in, hipStream_t stream) {
  ConvK k = k_in;
  k.pipe = 1;
  if (const char* env = getenv("LD_CONV_PIPE")) k.pipe = atoi(env) != 0;
  if constexpr (MODE != 2) {
    const int rc = launch_stream<MODE>(k, stream);
    if (rc != LD_EUNSUPPORTED) return rc;
  }
  const TileCfg c = pick_tile(k);
  const int nb = tile_blocks(k, c);
  // intra-block split-K when the grid would leave SIMDs with < ~4 waves
  int kgroups = 1;
  if (c.bm == 64 && c.bn == 64 && MODE != 2) {
    const int ksteps = ((k.Cin + c.bk - 1) / c.bk) * k.KH * k.KW;
    // measured (profiles/r01_kernels_s9.json): 2 groups pay off only on the
    // 25x42 stages (~1 workgroup per CU); neutral or negative elsewhere
    if (nb < 400 && ksteps >= 8) kgroups = 2;
  }
  if (const char* env = getenv("LD_CONV_KG")) {
    const int v = atoi(env);
    if (v == 1 || ((v == 2 || v == 4) && c.bm == 64 && c.bn == 64 && MODE != 2))
      kgroups = v;
  }
#define LD_CONV_LAUNCH(BM_, BN_, BK_, KG_)                                        \
  LD_LAUNCH((conv_igemm_kernel<BM_, BN_, BK_, MODE, KG_, 2, 2>),         \
                     dim3(nb), dim3(kThreads * KG_), 0, stream, k)
#define LD_CONV_CASE(BM_, BN_, BK_)                                               \
  if (c.bm == BM_ && c.bn == BN_ && c.bk == BK_) {                                \
    LD_CONV_LAUNCH(BM_, BN_, BK_, 1);                                             \
    return (int)hipGetLastError();                                                \
  }
  if (c.bm == 64 && c.bn == 64 && kgroups > 1) {
    if (c.bk == 32) {
      if (kgroups == 2) LD_CONV_LAUNCH(64, 64, 32, 2);
      else LD_CONV_LAUNCH(64, 64, 16, 4);  // 4 groups: 16-deep slices (LDS)
    } else {
      if (kgroups == 2) LD_CONV_LAUNCH(64, 64, 16, 2);
      else LD_CONV_LAUNCH(64, 64, 16, 4);
    }
    return (int)hipGetLastError();
  }
  if (c.bm == 32 && c.bn == 64) {  // 2-wavefront workgroups
    if (c.bk == 32)
      LD_LAUNCH((conv_igemm_kernel<32, 64, 32, MODE, 1, 1, 2>), dim3(nb),
                         dim3(128), 0, stream, k);
    else
      LD_LAUNCH((conv_igemm_kernel<32, 64, 16, MODE, 1, 1, 2>), dim3(nb),
                         dim3(128), 0, stream, k);
    return (int)hipGetLastError();
  }
  LD_CONV_CASE(128, 128, 16)
  LD_CONV_CASE(128, 64, 32)
  LD_CONV_CASE(64, 128, 32)
  LD_CONV_CASE(64, 64, 16)
  LD_CONV_CASE(64, 64, 32)
#undef LD_CONV_LAUNCH
#undef LD_CONV_CASE
  return LD_EUNSUPPORTED;
}

}  // namespace

// ---- the tuning table -------------------------------------------------------
namespace {
struct TuneKeyHash {
  size_t operator()(const LdTuneKey& k) const {
    size_t h = 1469598103934665603ull;
    for (int i = 0; i < 18; ++i) h = (h ^ (size_t)(unsigned)k.v[i]) * 1099511628211ull;
    return h;
  }
};
struct TuneKeyEq {
  bool operator()(const LdTuneKey& a, const LdTuneKey& b) const {
    for (int i = 0; i < 18; ++i)
      if (a.v[i] != b.v[i]) return false;
    return true;
  }
};
std::mutex g_tune_mu;
std::unordered_map<LdTuneKey, LdTuneCfg, TuneKeyHash, TuneKeyEq> g_tune;
}  // namespace

bool ld_tune_lookup(const LdTuneKey& key, LdTuneCfg* out) {
  std::lock_guard<std::mutex> lock(g_tune_mu);
  auto it = g_tune.find(key);
  if (it == g_tune.end()) return false;
  *out = it->second;
  return true;
}

void ld_tune_store(const LdTuneKey& key, const LdTuneCfg& cfg) {
  std::lock_guard<std::mutex> lock(g_tune_mu);
  g_tune[key] = cfg;
}

// Text format: one record per line, "18 key ints  tm tn wvm d ks [cap]"; '#' starts a
// comment line.  Returns the number of records read, or LD_EINVAL.
extern "C" int ld_conv_tune_load(const char* path) {
  if (!path || !*path) return LD_EINVAL;
  FILE* f = fopen(path, "r");
  if (!f) return LD_EINVAL;
  int n = 0;
  char line[1024];
  while (fgets(line, sizeof(line), f)) {
    if (line[0] == '#' || line[0] == '\n') continue;
    LdTuneKey k;
    LdTuneCfg c;
    int pos = 0, adv = 0, got = 0;
    for (int i = 0; i < 18; ++i)
      if (sscanf(line + pos, "%d%n", &k.v[i], &adv) == 1) {
        pos += adv;
        ++got;
      }
    c.cap = 0;
    if (got == 18 && sscanf(line + pos, "%d %d %d %d %d %d", &c.tm, &c.tn, &c.wvm, &c.d,
                            &c.ks, &c.cap) >= 5) {
      ld_tune_store(k, c);
      ++n;
    }
  }
  fclose(f);
  return n;
}

extern "C" int ld_conv_tune_save(const char* path) {
  if (!path || !*path) return LD_EINVAL;
  FILE* f = fopen(path, "w");
  if (!f) return LD_EINVAL;
  std::lock_guard<std::mutex> lock(g_tune_mu);
  fprintf(f, "# ld_amd conv shape table: MODE Cin Cout KH KW stride pad J levels Hin0 "
             "Win0 ph pw relu res affine family 0 | tm tn wvm d ks cap\n");
  for (const auto& kv : g_tune) {
    for (int i = 0; i < 18; ++i) fprintf(f, "%d ", kv.first.v[i]);
    fprintf(f, " %d %d %d %d %d %d\n", kv.second.tm, kv.second.tn, kv.second.wvm,
            kv.second.d, kv.second.ks, kv.second.cap);
  }
  fclose(f);
  return (int)g_tune.size();
}

extern "C" int ld_conv_tune_clear(void) {
  std::lock_guard<std::mutex> lock(g_tune_mu);
  g_tune.clear();
  return 0;
}

extern "C" size_t ld_conv_weight_image_floats(int Cout, int Cin, int KH, int KW,
                                             int backward) {
  if (Cout < 1 || Cin < 1 || KH < 1 || KW < 1) return 0;
  return backward ? (size_t)KH * KW * kpad_rows(Cout) * Cin
                  : (size_t)KH * KW * kpad_rows(Cin) * Cout;
}

extern "C" int ld_conv_weight_transform(const float* w, int Cout, int Cin, int KH,
                                        int KW, float* wt_fwd, float* wt_bwd,
                                        ld_stream_t stream) {
  if (!w || Cout < 1 || Cin < 1 || KH < 1 || KW < 1 || (!wt_fwd && !wt_bwd))
    return LD_EINVAL;
  const int cip = kpad_rows(Cin), cop = kpad_rows(Cout);
  size_t total = 0;
  if (wt_fwd) total = (size_t)KH * KW * cip * Cout;
  if (wt_bwd) total = max(total, (size_t)KH * KW * cop * Cin);
  LD_LAUNCH(conv_weight_transform_kernel,
                     dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, w, Cout, Cin, KH * KW, cip, cop, wt_fwd,
                     wt_bwd);
  return (int)hipGetLastError();
}

extern "C" int ld_conv_weight_transform_batch(const ld_wt_job_t* jobs,
                                              const int32_t* block_job, int nblocks,
                                              ld_stream_t stream) {
  if (!jobs || !block_job || nblocks < 1) return LD_EINVAL;
  LD_LAUNCH(conv_weight_transform_batch_kernel, dim3(nblocks), dim3(256), 0,
                     (hipStream_t)stream, jobs, block_job);
  return (int)hipGetLastError();
}

namespace {
int build_forward(const ld_conv_t* c, const float* x, const void* wt_fwd,
                  const ld_conv_epilogue_t* ep, float* y, ConvK& k, int family = 0) {
  if (int e = check_conv(c)) return e;
  if (!x || !wt_fwd) return LD_EINVAL;
  if (!y && !(family >= 1 && ep && ep->y_c8)) return LD_EINVAL;
  k = ConvK{};
  k.x = x;
  k.wt = (const float*)wt_fwd;
  k.y = y;
  k.bias = ep ? ep->bias : nullptr;
  k.scale = ep ? ep->scale : nullptr;
  k.shift = ep ? ep->shift : nullptr;
  k.residual = ep ? ep->residual : nullptr;
  k.relu = ep ? ep->relu : 0;
  k.y_c8 = ep ? ep->y_c8 : nullptr;
  k.res_c8 = ep ? ep->residual_c8 : nullptr;
  k.y_raw = ep ? ep->y_raw : nullptr;
  k.raw_c8 = ep ? ep->y_raw_c8 : nullptr;
  if (k.y_raw && (k.bias || !y)) return LD_EINVAL;  // raw = acc: no bias, y needed
  // the raw result as a C8 image: the C8-operand kernels only (family 2)
  if (k.raw_c8 && (family != 2 || k.bias || k.y_raw || c->Cout % 8 != 0))
    return LD_EINVAL;
  if ((k.y_c8 || k.res_c8) && (family == 0 || c->Cout % 8 != 0)) return LD_EINVAL;
  if (k.res_c8 && k.residual) return LD_EINVAL;
  if ((k.scale == nullptr) != (k.shift == nullptr)) return LD_EINVAL;
  k.N = c->N; k.Cin = c->Cin; k.Cout = c->Cout; k.KH = c->KH; k.KW = c->KW;
  k.g.stride = c->stride; k.g.pad = c->pad; k.Pin = c->Pin; k.Pout = c->Pout;
  k.g.num_levels = c->num_levels;
  k.J = c->N * c->Pout;
  for (int l = 0; l < LD_MAX_LEVELS; ++l) k.g.lv[l] = c->lv[l];
  if (family >= 1) {
    // bf16 image: [tap][Cin16 / 8][Cout][8] bf16 = 4 floats per (8-block, co)
    k.Kpad = (c->Cin + 15) / 16 * 2;
    if (family == 2) {  // x is the bf16 C8 image: half the bytes per element
      if (c->Cin % 32 != 0) return LD_EUNSUPPORTED;
      k.x_c8 = 1;
      return set_extents(k, ((size_t)c->N * c->Cin * c->Pin + 1) / 2,
                         (size_t)c->KH * c->KW * k.Kpad * c->Cout * 4);
    }
    return set_extents(k, (size_t)c->N * c->Cin * c->Pin,
                       (size_t)c->KH * c->KW * k.Kpad * c->Cout * 4);
  }
  k.Kpad = kpad_rows(c->Cin);
  return set_extents(k, (size_t)c->N * c->Cin * c->Pin,
                     (size_t)c->KH * c->KW * k.Kpad * c->Cout);
}
}  // namespace

// bf16-MFMA forward (conv_bf16.hip): x, epilogue operands and y are fp32 exactly
// as in ld_conv_forward; wt_fwd is the bf16 image of
// ld_conv_bf16_weight_transform.  Cin must be a multiple of 16
// (LD_EUNSUPPORTED otherwise: use ld_conv_forward).
extern "C" int ld_conv_bf16_forward(const ld_conv_t* c, const float* x,
                                    const void* wt_fwd, const ld_conv_epilogue_t* ep,
                                    float* y, ld_stream_t stream) {
  ConvK k;
  if (int e = build_forward(c, x, wt_fwd, ep, y, k, 1)) return e;
  return ld_bf16_stream_launch(0, k, (hipStream_t)stream);
}

// Same with the activation operand as the bf16 channel-blocked image of
// ld_conv_to_c8 (Cin a multiple of 32).
extern "C" int ld_conv_bf16_forward_c8(const ld_conv_t* c, const void* x_c8,
                                       const void* wt_fwd,
                                       const ld_conv_epilogue_t* ep, float* y,
                                       ld_stream_t stream) {
  ConvK k;
  if (int e = build_forward(c, (const float*)x_c8, wt_fwd, ep, y, k, 2)) return e;
  return ld_bf16_stream_launch(0, k, (hipStream_t)stream);
}

extern "C" int ld_conv_bf16_tune_forward_c8(const ld_conv_t* c, const void* x_c8,
                                            const void* wt_fwd,
                                            const ld_conv_epilogue_t* ep, float* y,
                                            ld_stream_t stream) {
  ConvK k;
  if (int e = build_forward(c, (const float*)x_c8, wt_fwd, ep, y, k, 2)) return e;
  return ld_bf16_stream_tune(0, k, (hipStream_t)stream);
}

extern "C" int ld_conv_bf16_tune_forward(const ld_conv_t* c, const float* x,
                                         const void* wt_fwd,
                                         const ld_conv_epilogue_t* ep, float* y,
                                         ld_stream_t stream) {
  ConvK k;
  if (int e = build_forward(c, x, wt_fwd, ep, y, k, 1)) return e;
  return ld_bf16_stream_tune(0, k, (hipStream_t)stream);
}

extern "C" int ld_conv_forward(const ld_conv_t* c, const float* x,
                               const float* wt_fwd, const ld_conv_epilogue_t* ep,
                               float* y, ld_stream_t stream) {
  ConvK k;
  if (int e = build_forward(c, x, wt_fwd, ep, y, k)) return e;
  return launch_igemm<0>(k, (hipStream_t)stream);
}

extern "C" int ld_conv_tune_forward(const ld_conv_t* c, const float* x,
                                    const float* wt_fwd,
                                    const ld_conv_epilogue_t* ep, float* y,
                                    ld_stream_t stream) {
  ConvK k;
  if (int e = build_forward(c, x, wt_fwd, ep, y, k)) return e;
  return tune_stream<0>(k, (hipStream_t)stream);
}

// Small-Cin convolution (the 7x7 stem, Cin = 3): the reduction index is the
// flat (ci, kh, kw) triple; wt is the [Cin*KH*KW][Cout] image, i.e.
// ld_conv_weight_transform(w, Cout, Cin*KH*KW, 1, 1, wt, NULL).
extern "C" int ld_conv_forward_smallc(const ld_conv_t* c, const float* x,
                                      const float* wt, const ld_conv_epilogue_t* ep,
                                      float* y, ld_stream_t stream) {
  if (int e = check_conv(c)) return e;
  if (!x || !wt || !y) return LD_EINVAL;
  ConvK k{};
  k.x = x;
  k.wt = wt;
  k.y = y;
  k.bias = ep ? ep->bias : nullptr;
  k.scale = ep ? ep->scale : nullptr;
  k.shift = ep ? ep->shift : nullptr;
  k.residual = ep ? ep->residual : nullptr;
  k.relu = ep ? ep->relu : 0;
  if (ep && (ep->y_c8 || ep->residual_c8 || ep->y_raw)) return LD_EINVAL;
  if ((k.scale == nullptr) != (k.shift == nullptr)) return LD_EINVAL;
  k.N = c->N; k.Cin = c->Cin; k.Cout = c->Cout; k.KH = c->KH; k.KW = c->KW;
  k.g.stride = c->stride; k.g.pad = c->pad; k.Pin = c->Pin; k.Pout = c->Pout;
  k.g.num_levels = c->num_levels;
  k.J = c->N * c->Pout;
  for (int l = 0; l < LD_MAX_LEVELS; ++l) k.g.lv[l] = c->lv[l];
  k.Kpad = kpad_rows(c->Cin * c->KH * c->KW);
  if (int e = set_extents(k, (size_t)c->N * c->Cin * c->Pin,
                          (size_t)k.Kpad * c->Cout))
    return e;
  // streaming stem kernel: all of Cout in one wave tile (32 or 64 channels);
  // LD_CONV_STEM=0 keeps the round-2 LDS tiling
  static const bool stem_on = [] {
    const char* e = getenv("LD_CONV_STEM");
    return !(e && e[0] == '0');
  }();
  if (stem_on && (c->Cout == 64 || c->Cout == 32) && k.Kpad % 32 == 0 &&
      (size_t)c->N * c->Cout * c->Pout * 4 < (size_t)kOOB) {
    hipStream_t st = (hipStream_t)stream;
    // round 6: both operands through LDS for THE stem geometry (3 -> 64, 7 x 7,
    // stride 2); LD_CONV_STEM=stream: the gather kernel
    const char* sel = getenv("LD_CONV_STEM");
    if (c->Cout == 64 && c->Cin == 3 && c->KH == 7 && c->KW == 7 && c->stride == 2 &&
        c->num_levels == 1 && !(sel && sel[0] == 's')) {
      const int tw = (c->lv[0].Wout + 63) / 64;
      if (sel && sel[0] == '8') {  // A/B: eight-row tiles (78 vs 81 TFLOP/s measured)
        const int th = (c->lv[0].Hout + 7) / 8;
        LD_LAUNCH((conv_stem_lds_kernel<8, 3, 7, 7, 2>), dim3(tw * th * c->N), dim3(512), 0,
                  st, k, tw, th);
      } else {
        const int th = (c->lv[0].Hout + 3) / 4;
        LD_LAUNCH((conv_stem_lds_kernel<4, 3, 7, 7, 2>), dim3(tw * th * c->N), dim3(256), 0,
                  st, k, tw, th);
      }
      return (int)hipGetLastError();
    }
    // 2 x 2 wave tiles of 32: 64 channels x 64 positions per wave, 4 waves per
    // workgroup, 8 k-pairs in flight
    if (c->Cout == 64) {
      const int nb = (k.J + 4 * 64 - 1) / (4 * 64);
      LD_LAUNCH((conv_stem_kernel<2, 2, 8>), dim3(nb), dim3(256), 0, st, k);
    } else {
      const int nb = (k.J + 4 * 128 - 1) / (4 * 128);
      LD_LAUNCH((conv_stem_kernel<1, 4, 8>), dim3(nb), dim3(256), 0, st, k);
    }
    return (int)hipGetLastError();
  }
  return launch_igemm<2>(k, (hipStream_t)stream);
}

// dx = conv_transpose(dy): runs the same implicit GEMM with the roles of the
// channel dims swapped, the flipped-tap weight image and pad' = K - 1 - pad.
// Stride 2: the output positions are split into their four (row, col) parity
// classes; each class sees only the taps of matching parity (1, 2, 2 or 4 of
// the 9 for a 3x3), so no MFMA work is spent on the dilation zeros.
namespace {
// addend (round 5): optional (N, Cin, Pin) tensor added in the epilogue -- dx =
// conv_transpose(dy) + addend, addend == dx allowed (every element is read and
// written by the same lane).  It is what autograd would otherwise do in a
// separate elementwise launch wherever a tensor has two consumers (the block
// input of a residual block: conv1 and the identity path; an FPN level feeding
// both head towers).  The launch uses the shape-table record of the plain dgrad.
int dgrad_walk(const ld_conv_t* c, const float* dy, const void* wt_bwd, float* dx,
               ld_stream_t stream_, bool tune, int family = 0,
               const float* addend = nullptr) {
  if (int e = check_conv(c)) return e;
  if (!dy || !wt_bwd || !dx) return LD_EINVAL;
  if (c->KH != c->KW) return LD_EUNSUPPORTED;
  hipStream_t stream = (hipStream_t)stream_;
  bool all_known = true;
  ConvK k{};
  k.x = dy;
  k.wt = (const float*)wt_bwd;
  k.y = dx;
  k.bias = k.scale = k.shift = nullptr;
  k.residual = addend;
  k.tune_plain = 1;
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
  k.ph = k.pw = k.kh0 = k.kw0 = k.nth = k.ntw = k.ch0 = k.cw0 = 0;
  k.Pfull = c->Pin;
  for (int l = 0; l < LD_MAX_LEVELS; ++l) {
    k.fW[l] = c->lv[l].Win;
    k.foff[l] = c->lv[l].off_in;
  }
  k.Kpad = family >= 1 ? (c->Cout + 15) / 16 * 2 : kpad_rows(c->Cout);
  if (family == 2) {  // dy is the bf16 C8 image
    if (c->Cout % 32 != 0) return LD_EUNSUPPORTED;
    k.x_c8 = 1;
  }
  if (int e = set_extents(
          k, family == 2 ? ((size_t)c->N * c->Cout * c->Pout + 1) / 2
                         : (size_t)c->N * c->Cout * c->Pout,
          (size_t)c->KH * c->KW * k.Kpad * c->Cin * (family >= 1 ? 4 : 1)))
    return e;
  if (family >= 1 && c->Cout % 16 != 0) return LD_EUNSUPPORTED;
  auto run = [&](int mode, const ConvK& q) -> int {
    if (family >= 1)
      return tune ? ld_bf16_stream_tune(mode, q, stream)
                  : ld_bf16_stream_launch(mode, q, stream);
    if (mode == 1) return tune ? tune_stream<1>(q, stream) : launch_igemm<1>(q, stream);
    return tune ? tune_stream<0>(q, stream) : launch_igemm<0>(q, stream);
  };
  if (c->stride == 1) return run(0, k);

  const int padp = c->KH - 1 - c->pad;
  // a class no tap reaches (1x1 stride 2: every odd row / column) is all
  // zeros: clear dx up front, then launch only the classes that have taps
  bool any_empty = false;
  for (int ph = 0; ph < 2; ++ph) {
    const int k0 = ((ph - padp) % 2 + 2) % 2;
    if (k0 >= c->KH) any_empty = true;
  }
  if (any_empty && !tune) {
    // ... or, with an addend, start from the addend (nothing to do in place)
    const size_t bytes = (size_t)c->N * c->Cin * c->Pin * sizeof(float);
    hipError_t err = hipSuccess;
    if (!addend)
      err = ldrec::memset_async(dx, 0, bytes, stream);
    else if (addend != dx)
      err = ldrec::memcpy_d2d_async(dx, addend, bytes, stream);
    if (err) return (int)err;
  }
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      ConvK q = k;
      q.ph = ph;
      q.pw = pw;
      // taps with (ph - pad' + kh) even
      q.kh0 = ((ph - padp) % 2 + 2) % 2;
      q.kw0 = ((pw - padp) % 2 + 2) % 2;
      q.nth = q.kh0 < c->KH ? (c->KH - q.kh0 + 1) / 2 : 0;
      q.ntw = q.kw0 < c->KW ? (c->KW - q.kw0 + 1) / 2 : 0;
      q.ch0 = (ph - padp + q.kh0) / 2;  // exact: the numerator is even
      q.cw0 = (pw - padp + q.kw0) / 2;
      int pc = 0;
      bool empty = false;
      for (int l = 0; l < c->num_levels; ++l) {
        const int Hc = (c->lv[l].Hin - ph + 1) / 2, Wc = (c->lv[l].Win - pw + 1) / 2;
        q.g.lv[l].Hout = Hc;
        q.g.lv[l].Wout = Wc;
        q.g.lv[l].off_out = pc;
        if (Hc <= 0 || Wc <= 0) empty = true;
        pc += max(Hc, 0) * max(Wc, 0);
      }
      if (empty && c->num_levels > 1) return LD_EUNSUPPORTED;
      if (pc == 0) continue;
      q.Pout = pc;
      q.J = c->N * pc;
      if (q.nth * q.ntw == 0) continue;  // zero class, cleared above
      const int e = run(1, q);
      if (tune) {
        if (e < 0 || e > 1) return e;
        all_known = all_known && e == 1;
      } else if (e) {
        return e;
      }
    }
  return tune && all_known ? 1 : 0;
}
}  // namespace

extern "C" int ld_conv_dgrad(const ld_conv_t* c, const float* dy,
                             const float* wt_bwd, float* dx, ld_stream_t stream) {
  return dgrad_walk(c, dy, wt_bwd, dx, stream, false);
}

extern "C" int ld_conv_dgrad_acc(const ld_conv_t* c, const float* dy,
                                 const float* wt_bwd, const float* addend, float* dx,
                                 ld_stream_t stream) {
  return dgrad_walk(c, dy, wt_bwd, dx, stream, false, 0, addend);
}

extern "C" int ld_conv_bf16_dgrad_acc(const ld_conv_t* c, const float* dy,
                                      const void* wt_bwd, const float* addend,
                                      float* dx, ld_stream_t stream) {
  return dgrad_walk(c, dy, wt_bwd, dx, stream, false, 1, addend);
}

extern "C" int ld_conv_bf16_dgrad_c8_acc(const ld_conv_t* c, const void* dy_c8,
                                         const void* wt_bwd, const float* addend,
                                         float* dx, ld_stream_t stream) {
  return dgrad_walk(c, (const float*)dy_c8, wt_bwd, dx, stream, false, 2, addend);
}

extern "C" int ld_conv_tune_dgrad(const ld_conv_t* c, const float* dy,
                                  const float* wt_bwd, float* dx,
                                  ld_stream_t stream) {
  return dgrad_walk(c, dy, wt_bwd, dx, stream, true);
}

// bf16-MFMA data gradient: Cout must be a multiple of 16 (it is the reduction).
extern "C" int ld_conv_bf16_dgrad(const ld_conv_t* c, const float* dy,
                                  const void* wt_bwd, float* dx, ld_stream_t stream) {
  return dgrad_walk(c, dy, wt_bwd, dx, stream, false, 1);
}

extern "C" int ld_conv_bf16_tune_dgrad(const ld_conv_t* c, const float* dy,
                                       const void* wt_bwd, float* dx,
                                       ld_stream_t stream) {
  return dgrad_walk(c, dy, wt_bwd, dx, stream, true, 1);
}

// data gradient with dy as the bf16 C8 image (Cout a multiple of 32)
extern "C" int ld_conv_bf16_dgrad_c8(const ld_conv_t* c, const void* dy_c8,
                                     const void* wt_bwd, float* dx,
                                     ld_stream_t stream) {
  return dgrad_walk(c, (const float*)dy_c8, wt_bwd, dx, stream, false, 2);
}

extern "C" int ld_conv_bf16_tune_dgrad_c8(const ld_conv_t* c, const void* dy_c8,
                                          const void* wt_bwd, float* dx,
                                          ld_stream_t stream) {
  return dgrad_walk(c, (const float*)dy_c8, wt_bwd, dx, stream, true, 2);
}

// which wgrad kernel: 0 = 128x128 workgroup tiles (LDS shared by 4 waves),
// 32 / 16 = wave-private 64x64 tiles with that many j per step
static int wgrad_mode() {
  if (const char* env = getenv("LD_CONV_WGRAD")) {
    const int v = atoi(env);
    if (v == 0 || v == 16 || v == 32) return v;
  }
  return 32;
}

static int wgrad_splits(const ld_conv_t* c) {
  const int J = c->N * c->Pout;
  const int mode = wgrad_mode();
  const int edge = mode ? 64 : 128, bk = mode ? mode : WBK;
  const int tiles = ((c->Cout + edge - 1) / edge) * ((c->Cin + edge - 1) / edge) *
                    c->KH * c->KW;
  if (const char* env = getenv("LD_CONV_WGRAD_TARGET")) {
    const int target = max(1, atoi(env));
    int splits = (target + tiles - 1) / tiles;
    const int max_by_k = (J + 8 * bk - 1) / (8 * bk);  // >= 8 steps per block
    return max(1, min(min(splits, max_by_k), 64));
  }
  if (!mode) {
    int splits = (1024 + tiles - 1) / tiles;            // aim at ~1024 workgroups
    const int max_by_k = (J + 8 * bk - 1) / (8 * bk);
    return max(1, min(min(splits, max_by_k), 64));
  }
  // Wave-private kernel: every wavefront is a workgroup and the device holds
  // `slots` of them at once (VGPR-limited: 2 per SIMD at 32 j/step, 3 at 16).
  // A launch runs in ceil(waves / slots) rounds of (steps + fixed) each, so the
  // split count is chosen to fill whole rounds -- 2112 waves on 2048 slots cost
  // two rounds, 2048 cost one (profiles/r01_kernels_s24_wgrad.txt).
  const int slots = 1024 * (mode == 32 ? 2 : 3);
  const double fixed = mode == 32 ? 5.0 : 8.0;  // prologue + 64x64 slab store, in steps
  int best = 1;
  double best_cost = 0;
  const size_t wbytes = (size_t)c->KH * c->KW * c->Cout * c->Cin * sizeof(float);
  for (int sp = 1; sp <= 256; ++sp) {
    if (sp > 1 && sp * wbytes > ((size_t)256 << 20)) break;  // slab budget
    int jchunk = (J + sp - 1) / sp;
    jchunk = (jchunk + bk - 1) / bk * bk;
    if (sp > 1 && (long)(sp - 1) * jchunk >= J) continue;  // empty last split
    const int steps = jchunk / bk;
    if (sp > 1 && steps < 4) break;
    const long waves = (long)tiles * sp;
    const double rounds = (double)((waves + slots - 1) / slots);
    // slab traffic of the reduce pass, in step units (rough: 64x64 floats
    // written + read per wave vs 16 KB of operand tile per step)
    const double cost = rounds * (steps + fixed) + 1e-3 * sp;
    if (sp == 1 || cost < best_cost) {
      best = sp;
      best_cost = cost;
    }
  }
  return best;
}

// ---- which fp32 weight-gradient kernel, and how it is split (round 4) ---------
// kind 1 = conv_wgrad_tile_kernel (conv_wgrad.hip: 128 x 128 workgroup tiles, kg
// k-groups, bk columns per slice, `splits` workgroups per tile, fused = splits
// combined inside the launch), kind 0 = the wave-private kernel above with its
// own split model.  Order: LD_CONV_WGRAD_CFG="kind,kg,bk,splits,fused" (tools),
// the tuning table (key MODE 2, family 0; ld_conv_tune_wgrad), else a model
// that is a pure function of the geometry -- all ranks pick the same.
struct WgCfg {
  int kind, kg, bk, splits, fused;
};

static LdTuneKey wgrad_tune_key(const ld_conv_t* c) {
  return LdTuneKey{{2, c->Cin, c->Cout, c->KH, c->KW, c->stride, c->pad, c->N * c->Pout,
                    c->num_levels, c->lv[0].Hin, c->lv[0].Win, 0, 0, 0, 0, 0, 0, 0}};
}

// most splits a tile config may use: >= 2 slices per workgroup, <= 128 MB of partials
static int wgrad_tile_max_splits(const ld_conv_t* c, int bk) {
  const int J = c->N * c->Pout;
  const int ntaps = c->KH * c->KW;
  int by_k = J / (2 * bk);
  if (by_k < 1) by_k = 1;
  int sp = by_k < 256 ? by_k : 256;
  while (sp > 1 && ld_f32_wgrad_tile_workspace(c->Cout, c->Cin, ntaps, sp) > ((size_t)128 << 20))
    --sp;
  return sp;
}

static WgCfg wgrad_model(const ld_conv_t* c) {
  // Untuned geometries; the rules are the pattern of the round-4 sweep on the C2
  // shapes (profiles/r04_wgrad_sweep.json).  In-launch combination of the splits
  // (fused) measured slower than the slab reduce launch everywhere: off.
  WgCfg g{0, 0, 0, 0, 0};
  if (c->Cout < 48 || c->Cin < 48) return g;  // stem-like: wave-private 64 x 64 tiles
  const int J = c->N * c->Pout;
  const int ntaps = c->KH * c->KW;
  if (c->KH == 3 && c->KW == 3) {
    // three kw taps per workgroup when a workgroup still gets >= ~1000 columns
    const int tiles3 = ((c->Cout + 127) / 128) * ((c->Cin + 127) / 128) * 3;
    if ((long)J * tiles3 >= 256L * 1000) {
      g.kind = 2;
      int sp = 256 / tiles3;
      if (sp < 1) sp = 1;
      const int mx = wgrad_tile_max_splits(c, 32);
      g.splits = sp < mx ? sp : mx;
      return g;
    }
  }
  g.kind = 1;
  g.kg = 2;
  g.bk = 64;
  const int tiles = ((c->Cout + 127) / 128) * ((c->Cin + 127) / 128) * ntaps;
  const int slots = ld_f32_wgrad_tile_slots(g.kg, g.bk);
  int sp = slots / tiles;
  if (sp < 1) sp = 1;
  const int mx = wgrad_tile_max_splits(c, g.bk);
  g.splits = sp < mx ? sp : mx;
  return g;
}

static WgCfg wgrad_pick(const ld_conv_t* c) {
  if (const char* env = getenv("LD_CONV_WGRAD_CFG")) {
    WgCfg g{0, 0, 0, 0, 0};
    if (sscanf(env, "%d,%d,%d,%d,%d", &g.kind, &g.kg, &g.bk, &g.splits, &g.fused) >= 1) {
      if (g.kind == 0) return g;
      if (g.kind == 2 && c->KH == 3 && c->KW == 3) {
        const int mx = wgrad_tile_max_splits(c, 32);
        g.splits = g.splits < 1 ? 1 : (g.splits > mx ? mx : g.splits);
        return g;
      }
      if (ld_f32_wgrad_tile_cfg_ok(g.kg, g.bk)) {
        const int mx = wgrad_tile_max_splits(c, g.bk);
        if (g.splits < 1) g.splits = 1;
        if (g.splits > mx) g.splits = mx;
        return g;
      }
    }
  }
  LdTuneCfg t;
  if (ld_tune_lookup(wgrad_tune_key(c), &t)) {
    WgCfg g{t.tm, t.tn, t.wvm, t.d, t.ks};
    if (g.kind == 0) return g;
    if (g.kind == 2 && c->KH == 3 && c->KW == 3) {
      const int mx = wgrad_tile_max_splits(c, 32);
      g.splits = g.splits < 1 ? 1 : (g.splits > mx ? mx : g.splits);
      return g;
    }
    if (ld_f32_wgrad_tile_cfg_ok(g.kg, g.bk)) {
      const int mx = wgrad_tile_max_splits(c, g.bk);
      if (g.splits < 1) g.splits = 1;
      if (g.splits > mx) g.splits = mx;
      return g;
    }
  }
  return wgrad_model(c);
}

// Workspace of the plan a launch of this geometry ACTUALLY uses (round 4
// returned the tuner's worst case for every family, ~250 MB of device memory more
// than the shipped plans need): per kernel family the split count that family's
// launch picks; family 0 = ld_conv_wgrad (the record of the shape table, else the
// model), 1 = ld_conv_bf16_wgrad, 2 = ld_conv_bf16_wgrad_c8.
static size_t wgrad_need(const ld_conv_t* c, int family, const WgCfg* forced) {
  const int ntaps = c->KH * c->KW;
  const size_t per = (size_t)ntaps * c->Cout * c->Cin * sizeof(float);
  if (family == 2 && ld_bf16_wgrad_c8_tiled(c->Cout, c->Cin))
    return per * ld_bf16_wgrad_c8_tile_splits(c->Cout, c->Cin, ntaps, c->N * c->Pout);
  if (family == 1 && ld_bf16_wgrad_tiled(c->Cout, c->Cin, c->Pout))
    return per * ld_bf16_wgrad_splits(c->Cout, c->Cin, ntaps, c->N * c->Pout);
  if (family >= 1) return per * wgrad_splits(c);
  const WgCfg g = forced ? *forced : wgrad_pick(c);
  if (g.kind == 0) return per * wgrad_splits(c);
  const size_t tile = ld_f32_wgrad_tile_workspace(c->Cout, c->Cin, ntaps, g.splits);
  return tile > per ? tile : per;
}

// One size that serves whichever of the three launch entry points the caller
// takes for this geometry (the host keeps one buffer per stream).
extern "C" size_t ld_conv_wgrad_workspace_bytes(const ld_conv_t* c) {
  if (check_conv(c) != 0) return 0;
  size_t need = wgrad_need(c, 0, nullptr);
  if (c->Cin >= 16) {
    const size_t b = wgrad_need(c, 1, nullptr);
    need = b > need ? b : need;
  }
  if (c->Cin % 8 == 0 && c->Cout % 8 == 0) {
    const size_t b = wgrad_need(c, 2, nullptr);
    need = b > need ? b : need;
  }
  return need;
}

// ld_conv_tune_wgrad times every candidate plan on the caller's workspace: the
// worst case over them (capped at 128 MB by wgrad_tile_max_splits).
extern "C" size_t ld_conv_tune_wgrad_workspace_bytes(const ld_conv_t* c) {
  if (check_conv(c) != 0) return 0;
  const size_t need = ld_conv_wgrad_workspace_bytes(c);
  const size_t tile = ld_f32_wgrad_tile_workspace(c->Cout, c->Cin, c->KH * c->KW,
                                                  wgrad_tile_max_splits(c, 32));
  return need > tile ? need : tile;
}

namespace {
// defer != nullptr (round 5): only the split partials are computed, as slabs in
// `workspace` (which the caller keeps until ld_wgrad_reduce_batch has run), and
// *defer receives the reduction job; dw / accumulate are not touched.
int wgrad_run(const ld_conv_t* c, const float* x, const float* dy, float* dw,
              int accumulate, void* workspace, size_t workspace_bytes,
              ld_stream_t stream_, int family, const WgCfg* forced = nullptr,
              ld_wgrad_job_t* defer = nullptr) {
  if (int e = check_conv(c)) return e;
  if (!x || !dy || (!dw && !defer)) return LD_EINVAL;
  if (!workspace || workspace_bytes < wgrad_need(c, family, forced)) return LD_ENOSPACE;
  hipStream_t stream = (hipStream_t)stream_;
  WgradK k;
  k.x = x; k.dy = dy; k.slabs = (float*)workspace;
  k.N = c->N; k.Cin = c->Cin; k.Cout = c->Cout; k.KH = c->KH; k.KW = c->KW;
  k.g.stride = c->stride; k.g.pad = c->pad; k.Pin = c->Pin; k.Pout = c->Pout;
  k.g.num_levels = c->num_levels;
  k.J = c->N * c->Pout;
  k.splits = wgrad_splits(c);
  if (family == 1 && ld_bf16_wgrad_tiled(c->Cout, c->Cin, c->Pout))
    k.splits = ld_bf16_wgrad_splits(c->Cout, c->Cin, c->KH * c->KW, k.J);
  if (family == 2 && ld_bf16_wgrad_c8_tiled(c->Cout, c->Cin))
    k.splits = ld_bf16_wgrad_c8_tile_splits(c->Cout, c->Cin, c->KH * c->KW, k.J);
  const int wmode = wgrad_mode();
  const int wbk = family >= 1 ? 32 : (wmode ? wmode : WBK);
  int jchunk = (k.J + k.splits - 1) / k.splits;
  jchunk = (jchunk + wbk - 1) / wbk * wbk;
  k.jchunk = jchunk;
  for (int l = 0; l < LD_MAX_LEVELS; ++l) k.g.lv[l] = c->lv[l];
  {
    const size_t xf = (size_t)c->N * c->Cin * c->Pin;
    const size_t yf = (size_t)c->N * c->Cout * c->Pout;
    if (xf * 4 >= (size_t)kOOB || yf * 4 >= (size_t)kOOB) return LD_EUNSUPPORTED;
    k.x_bytes = (unsigned)(xf * 4);
    k.dy_bytes = (unsigned)(yf * 4);
  }
  const int ntaps = c->KH * c->KW;
  int nslabs = k.splits;
  auto deferred = [&]() -> int {
    defer->slabs = (const float*)workspace;
    defer->splits = nslabs;
    defer->ntaps = ntaps;
    defer->Cout = c->Cout;
    defer->Cin = c->Cin;
    return 0;
  };
  if (family == 2) {  // x / dy are bf16 C8 images: half the bytes
    k.x_bytes = (unsigned)((size_t)c->N * c->Cin * c->Pin * 2);
    k.dy_bytes = (unsigned)((size_t)c->N * c->Cout * c->Pout * 2);
    if (int e = ld_bf16_wgrad_c8_launch(k, stream)) return e;
  } else if (family == 1) {
    if (int e = ld_bf16_wgrad_launch(k, stream)) return e;
  } else if (const WgCfg cfg = forced ? *forced : wgrad_pick(c); cfg.kind == 1) {
    const int e = ld_f32_wgrad_tile_launch(k, cfg.kg, cfg.bk, cfg.splits, cfg.fused, dw,
                                           accumulate, workspace, workspace_bytes, stream,
                                           defer ? &nslabs : nullptr);
    return e || !defer ? e : deferred();
  } else if (cfg.kind == 2) {
    const int e = ld_f32_wgrad_tap3_launch(k, cfg.splits, dw, accumulate, workspace,
                                           workspace_bytes, stream,
                                           defer ? &nslabs : nullptr);
    return e || !defer ? e : deferred();
  } else if (wmode) {
    const int blocks = ((c->Cout + 63) / 64) * ((c->Cin + 63) / 64) * ntaps * k.splits;
    if (wmode == 32)
      LD_LAUNCH(conv_wgrad_wave_kernel<32>, dim3(blocks), dim3(64), 0,
                         stream, k);
    else
      LD_LAUNCH(conv_wgrad_wave_kernel<16>, dim3(blocks), dim3(64), 0,
                         stream, k);
  } else {
    const int blocks =
        ((c->Cout + 127) / 128) * ((c->Cin + 127) / 128) * ntaps * k.splits;
    LD_LAUNCH(conv_wgrad_kernel, dim3(blocks), dim3(kThreads), 0, stream, k);
  }
  if (defer) {
    if (hipError_t e = hipGetLastError()) return (int)e;
    return deferred();
  }
  return ld_wgrad_reduce_launch(k.slabs, k.splits, ntaps, c->Cout, c->Cin, dw,
                                accumulate, stream);
}

// dW (+)= sum_split slab, every job of the table in ONE launch: block b serves
// 1024 consecutive [tap][co][ci] elements of job block_job[b].  Per element the
// splits are added in index order, exactly as conv_wgrad_reduce(4)_kernel does.
__global__ __launch_bounds__(256) void conv_wgrad_reduce_batch_kernel(
    const ld_wgrad_job_t* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  const ld_wgrad_job_t j = jobs[block_job[blockIdx.x]];
  const size_t per = (size_t)j.ntaps * j.Cout * j.Cin;
  const size_t i = ((size_t)(blockIdx.x - j.first_block) * 256 + threadIdx.x) * 4;
  if (i >= per) return;
  const bool vec = j.Cin % 4 == 0 && ((uintptr_t)j.slabs & 15) == 0 &&
                   (j.ntaps > 1 || ((uintptr_t)j.dw & 15) == 0);
  if (vec) {
    const int ci = (int)(i % j.Cin);
    const size_t q = i / j.Cin;
    const int co = (int)(q % j.Cout), tap = (int)(q / j.Cout);
    floatx4 s = {0.0f, 0.0f, 0.0f, 0.0f};
    int k = 0;
    for (; k + 8 <= j.splits; k += 8) {
      floatx4 v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u)
        v[u] = *reinterpret_cast<const floatx4*>(j.slabs + (size_t)(k + u) * per + i);
#pragma unroll
      for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; k < j.splits; ++k)
      s += *reinterpret_cast<const floatx4*>(j.slabs + (size_t)k * per + i);
    if (j.ntaps == 1) {
      float* o = j.dw + (size_t)co * j.Cin + ci;
      if (j.accumulate) s += *reinterpret_cast<const floatx4*>(o);
      *reinterpret_cast<floatx4*>(o) = s;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const size_t o = ((size_t)co * j.Cin + ci + e) * j.ntaps + tap;
        j.dw[o] = j.accumulate ? j.dw[o] + s[e] : s[e];
      }
    }
    return;
  }
  for (int e = 0; e < 4 && i + e < per; ++e) {
    const size_t ie = i + e;
    const int ci = (int)(ie % j.Cin);
    const size_t q = ie / j.Cin;
    const int co = (int)(q % j.Cout), tap = (int)(q / j.Cout);
    float s = 0.0f;
    for (int k = 0; k < j.splits; ++k) s += j.slabs[(size_t)k * per + ie];
    const size_t o = ((size_t)co * j.Cin + ci) * j.ntaps + tap;
    j.dw[o] = j.accumulate ? j.dw[o] + s : s;
  }
}
}  // namespace

extern "C" int ld_conv_wgrad_partial(const ld_conv_t* c, int family, const void* x,
                                     const void* dy, void* slabs, size_t slab_bytes,
                                     ld_wgrad_job_t* job, ld_stream_t stream) {
  if (!job || family < 0 || family > 2) return LD_EINVAL;
  if (family == 2 && c && (c->Cin % 8 != 0 || c->Cout % 8 != 0)) return LD_EUNSUPPORTED;
  return wgrad_run(c, (const float*)x, (const float*)dy, nullptr, 0, slabs, slab_bytes,
                   stream, family, nullptr, job);
}

extern "C" int ld_wgrad_reduce_batch(const ld_wgrad_job_t* jobs, const int32_t* block_job,
                                     int nblocks, ld_stream_t stream) {
  if (!jobs || !block_job || nblocks < 1) return LD_EINVAL;
  LD_LAUNCH(conv_wgrad_reduce_batch_kernel, dim3(nblocks), dim3(256), 0,
                     (hipStream_t)stream, jobs, block_job);
  return (int)hipGetLastError();
}

int ld_wgrad_reduce_launch(const float* slabs, int splits, int ntaps, int Cout, int Cin,
                           float* dw, int accumulate, hipStream_t stream) {
  const size_t per = (size_t)ntaps * Cout * Cin;
  if (Cin % 4 == 0 && (uintptr_t)slabs % 16 == 0 &&
      (ntaps > 1 || (uintptr_t)dw % 16 == 0))  // 1x1: dW rows are stored 16 bytes at a time
    LD_LAUNCH(conv_wgrad_reduce4_kernel,
                       dim3((unsigned)((per / 4 + 255) / 256)), dim3(256), 0, stream,
                       slabs, splits, ntaps, Cout, Cin, dw, accumulate);
  else
    LD_LAUNCH(conv_wgrad_reduce_kernel, dim3((unsigned)((per + 255) / 256)),
                       dim3(256), 0, stream, slabs, splits, ntaps, Cout, Cin, dw,
                       accumulate);
  return (int)hipGetLastError();
}

extern "C" int ld_conv_wgrad(const ld_conv_t* c, const float* x, const float* dy,
                             float* dw, int accumulate, void* workspace,
                             size_t workspace_bytes, ld_stream_t stream) {
  return wgrad_run(c, x, dy, dw, accumulate, workspace, workspace_bytes, stream, 0);
}

// Which kernel / split a launch of ld_conv_wgrad would use for this geometry
// (host logic only, no device work): out = {kind, kg, bk, splits, fused}.
extern "C" int ld_conv_wgrad_plan(const ld_conv_t* c, int* out) {
  if (int e = check_conv(c)) return e;
  if (!out) return LD_EINVAL;
  const WgCfg g = wgrad_pick(c);
  out[0] = g.kind;
  out[1] = g.kg;
  out[2] = g.bk;
  out[3] = g.kind == 0 ? wgrad_splits(c) : g.splits;
  out[4] = g.fused;
  return 0;
}

// Explicit tuning of the fp32 weight gradient (the counterpart of
// ld_conv_tune_forward): times the wave-private kernel and every (kg, bk, splits,
// fused) instance of the workgroup-tiled kernel that fits the geometry on the
// caller's buffers (dw is overwritten with the same result by every candidate)
// and records the winner.  Synchronises; not on a capturing stream.
extern "C" int ld_conv_tune_wgrad(const ld_conv_t* c, const float* x, const float* dy,
                                  float* dw, void* workspace, size_t workspace_bytes,
                                  ld_stream_t stream_) {
  if (int e = check_conv(c)) return e;
  if (!x || !dy || !dw || !workspace) return LD_EINVAL;
  if (workspace_bytes < ld_conv_tune_wgrad_workspace_bytes(c)) return LD_ENOSPACE;
  const LdTuneKey key = wgrad_tune_key(c);
  LdTuneCfg have;
  if (ld_tune_lookup(key, &have)) return 1;
  hipStream_t stream = (hipStream_t)stream_;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(stream, &cap);
  if (cap != hipStreamCaptureStatusNone) return LD_EUNSUPPORTED;
  (void)hipDeviceSynchronize();
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  // candidates: the wave-private kernel, then the tile instances
  std::vector<WgCfg> cands;
  cands.push_back(WgCfg{0, 0, 0, 0, 0});
  if (c->Cout >= 48 && c->Cin >= 48) {
    static const int kShapes[][2] = {{1, 32}, {2, 32}, {4, 32}, {2, 64}, {4, 64}};
    const int ntaps = c->KH * c->KW;
    const int tiles = ((c->Cout + 127) / 128) * ((c->Cin + 127) / 128) * ntaps;
    for (const auto& sh : kShapes) {
      const int slots = ld_f32_wgrad_tile_slots(sh[0], sh[1]);
      const int mx = wgrad_tile_max_splits(c, sh[1]);
      int last = 0;
      // whole rounds of the device and the steps between them
      const int tries[] = {slots / tiles, (slots + tiles - 1) / tiles, slots / (2 * tiles),
                           (3 * slots) / (4 * tiles), 2 * slots / tiles, 1};
      std::vector<int> seen;
      for (int sp : tries) {
        if (sp < 1) sp = 1;
        if (sp > mx) sp = mx;
        bool dup = false;
        for (int v : seen) dup |= v == sp;
        if (dup) continue;
        seen.push_back(sp);
        cands.push_back(WgCfg{1, sh[0], sh[1], sp, 0});
        if (sp > 1) cands.push_back(WgCfg{1, sh[0], sh[1], sp, 1});
      }
      (void)last;
    }
  }
  if (c->KH == 3 && c->KW == 3 && c->Cout >= 48 && c->Cin >= 48) {
    // the three kw taps of a kernel row per workgroup, one workgroup per CU
    const int tiles3 = ((c->Cout + 127) / 128) * ((c->Cin + 127) / 128) * 3;
    const int mx = wgrad_tile_max_splits(c, 32);
    std::vector<int> seen;
    const int tries[] = {256 / tiles3, (256 + tiles3 - 1) / tiles3, 128 / tiles3,
                         192 / tiles3, 384 / tiles3, 512 / tiles3, 1};
    for (int sp : tries) {
      if (sp < 1) sp = 1;
      if (sp > mx) sp = mx;
      bool dup = false;
      for (int v : seen) dup |= v == sp;
      if (dup) continue;
      seen.push_back(sp);
      cands.push_back(WgCfg{2, 0, 0, sp, 0});
    }
  }
  float best_ms = -1.0f;
  WgCfg best = cands[0];
  for (const WgCfg& g : cands) {
    if (wgrad_run(c, x, dy, dw, 0, workspace, workspace_bytes, stream_, 0, &g) != 0) continue;
    float ms = -1.0f;
    for (int trial = 0; trial < 2; ++trial) {
      (void)hipEventRecord(e0, stream);
      for (int rep = 0; rep < kTuneReps; ++rep)
        wgrad_run(c, x, dy, dw, 0, workspace, workspace_bytes, stream_, 0, &g);
      (void)hipEventRecord(e1, stream);
      if (hipEventSynchronize(e1) != hipSuccess) break;
      float t = 0.0f;
      (void)hipEventElapsedTime(&t, e0, e1);
      if (ms < 0.0f || t < ms) ms = t;
    }
    if (const char* lg = getenv("LD_CONV_TUNE_LOG"))
      if (lg[0] == '2')
        fprintf(stderr, "[ld_conv]   wgrad cand %d,%d,%d,%d,%d  %.1f us\n", g.kind, g.kg, g.bk,
                g.splits, g.fused, ms * 1e3 / kTuneReps);
    if (ms >= 0.0f && (best_ms < 0.0f || ms < best_ms)) {
      best_ms = ms;
      best = g;
    }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (const char* lg = getenv("LD_CONV_TUNE_LOG"))
    if (lg[0] == '1' || lg[0] == '2') {
      const double fl = 2.0 * c->N * c->Pout * (double)c->Cout * c->Cin * c->KH * c->KW;
      fprintf(stderr,
              "[ld_conv] wgrad Cin %d Cout %d k %dx%d s%d J %d lv %d -> kind %d kg %d bk %d "
              "splits %d fused %d  %.1f TFLOP/s\n",
              c->Cin, c->Cout, c->KH, c->KW, c->stride, c->N * c->Pout, c->num_levels,
              best.kind, best.kg, best.bk, best.splits, best.fused,
              best_ms > 0 ? fl / (best_ms * 1e-3 / kTuneReps) / 1e12 : 0.0);
    }
  if (best_ms > 0.0f)
    ld_tune_store(key, LdTuneCfg{best.kind, best.kg, best.bk, best.splits, best.fused, 0});
  return 0;
}

// bf16-MFMA weight gradient: fp32 x / dy / dw, operands rounded to bf16 on the
// way into the matrix core, fp32 slabs summed in fixed order.  Same workspace
// as ld_conv_wgrad.
// bf16 weight gradient with BOTH operands as C8 images (ld_conv_to_c8 layout;
// Cin and Cout multiples of 8)
extern "C" int ld_conv_bf16_wgrad_c8(const ld_conv_t* c, const void* x_c8,
                                     const void* dy_c8, float* dw, int accumulate,
                                     void* workspace, size_t workspace_bytes,
                                     ld_stream_t stream) {
  if (c && (c->Cin % 8 != 0 || c->Cout % 8 != 0)) return LD_EUNSUPPORTED;
  return wgrad_run(c, (const float*)x_c8, (const float*)dy_c8, dw, accumulate,
                   workspace, workspace_bytes, stream, 2);
}

extern "C" int ld_conv_bf16_wgrad(const ld_conv_t* c, const float* x, const float* dy,
                                  float* dw, int accumulate, void* workspace,
                                  size_t workspace_bytes, ld_stream_t stream) {
  return wgrad_run(c, x, dy, dw, accumulate, workspace, workspace_bytes, stream, 1);
}
