// bf16 convolution, C8 operands, 8-wave workgroup tiles fed by LDS-DMA (round 6).
//
// Same implicit GEMM, operands and accumulation order as conv_tile_c8_kernel in
// conv_bf16.hip (Y[Cout][J] = sum_k Wt[Cout][k] X[k][J], k = (tap, ci), bf16
// operands, fp32 accumulate, one accumulator per output element, k ascending) --
// results are bit-identical to it -- on a different machine mapping:
//   * workgroup = 8 wavefronts (512 threads), one workgroup per CU, tile BM x BN
//     up to 256 x 256; a wave owns (BM / WM) x (BN / WN), e.g. 128 x 64 = 4 x 2
//     accumulator tiles of 32 x 32 (128 accumulator registers): 6 fragment reads
//     per 8 MFMAs instead of the 4-wave kernel's 4 per 4;
//   * operands go global -> LDS directly (`buffer_load_dwordx4 ... lds`): no
//     register ring, no ds_write.  The LDS image of a 64-deep k-step is
//     [k8 = 8][rows][8 bf16]; a wave instruction fills 64 consecutive rows of
//     one k8 group = 1 KiB, lane-linear.  Both source images are already in that
//     order: the weight image is [tap][K/8][Cout][8] and the C8 activation image
//     (N, C/8, P, 8), so the per-lane SOURCE address carries everything (tap
//     shift, padding as an out-of-range offset -> the DMA writes zeros) and no
//     swizzle is needed: a fragment is 32 consecutive 16-byte rows per half-wave,
//     which ds_read_b128 serves without bank conflicts.
//   * two LDS stages of 64 KB (256 x 256).  A step = four 16-deep sub-steps, each
//     multiplying the fragments read during the previous one.  ONE barrier per
//     step, between sub-steps 2 and 3: by then the wave's reads of the current stage
//     are complete (lgkmcnt) and its share of the next step has landed (vmcnt), so
//     the fourth sub-step runs over the reads of the next step's first fragments;
//     the DMA of step u + 2 into the stage everybody just left is issued in three
//     groups (behind the barrier, then under the next step's first two sub-steps)
//     -- a burst of all 56 wave instructions of a CU queues on its address unit.
//   * epilogue: fp32 outputs with EXCHANGED operand roles (SWAP, see the kernel):
//     a lane then holds four consecutive positions of a channel; each 32 x 32 tile
//     takes a turn through a wave-private LDS buffer so that a wave instruction
//     stores eight whole 128-byte rows.  C8 outputs / C8 residuals / the stride-2
//     data-gradient classes keep the 4-wave kernel's epilogue.
// Measured (profiles/r06_t256_stamps.txt, r06_t256_head.json): head tower 256 -> 256
// 3 x 3, J = 44 800: 680 -> 890 TFLOP/s; main loop 72 % MFMA-busy (the DMA costs 23 %
// of it), epilogue at the HBM write rate, 234 tiles on 256 CUs.
// Reference op: the conv layers of mmdet/models/dense_heads/gfl_head.py:102-133
// (towers), necks/fpn.py:66-221, backbones/resnet.py:260-299, in the fp16 mode
// mmcv auto_fp16 gives them.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// 16 bytes per lane, global -> LDS at (wave-uniform dst) + lane * 16
__device__ __forceinline__ void dma16(rsrc_t r, uintx4* dst, unsigned voff,
                                      unsigned soff) {
#if defined(LD_T256_ABL) && (LD_T256_ABL & 4)  // ablation: the same loads into registers
  if (voff != 0x7fffff00u) {
    uintx4 v = __builtin_bit_cast(uintx4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    asm volatile("" ::"v"(v));
    return;
  }
#endif
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)dst, 16, voff, soff, 0, 0);
}

// Debug build only (LD_BUILD_DEFS=-DLD_T256_STAMP): lane 0 of every wave writes
// the shader clock at five points of the kernel into a buffer set through
// ld_debug_t256_stamps (tools/t256_stamps.py prints the phase breakdown).
#ifndef LD_T256_SPREAD
#define LD_T256_SPREAD 1  // 0: the whole DMA of a step in one burst (A/B builds)
#endif
#ifdef LD_T256_STAMP
__device__ unsigned long long* g_t256_stamps;
#define LD_STAMP(i)                                                               \
  do {                                                                            \
    if (lane == 0 && g_t256_stamps)                                               \
      g_t256_stamps[((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = clock64();       \
  } while (0)
#define LD_STAMP_WALL(i)                                                          \
  do {                                                                            \
    if (lane == 0 && g_t256_stamps)                                               \
      g_t256_stamps[((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = wall_clock64();   \
  } while (0)
#define LD_STAMP_VAL(i, v)                                                        \
  do {                                                                            \
    if (lane == 0 && g_t256_stamps)                                               \
      g_t256_stamps[((size_t)blockIdx.x * 8 + wave) * 16 + (i)] = (v);             \
  } while (0)
#else
#define LD_STAMP(i)
#define LD_STAMP_WALL(i)
#define LD_STAMP_VAL(i, v)
#endif

// How the 8 waves split the 8 x (rows / 64) wave-instructions of one operand
// image: when the chunk count divides 8 a wave keeps ONE 64-row chunk (so a lane
// keeps one row / position for the whole kernel) and walks C of the k8 groups;
// otherwise (192 rows) wave w takes k8 group w and walks the chunks.
template <int C>
struct LoadMap {
  static constexpr bool fixed = (8 % C) == 0;
  static constexpr int NPOS = fixed ? 1 : C;  // rows a lane addresses
  static constexpr int NI = C;                // wave-instructions per step
  __device__ static int chunk(int w, int i) { return fixed ? w % C : i; }
  __device__ static int kg(int w, int i) { return fixed ? (w / C) * C + i : w; }
  __device__ static int pos(int i) { return fixed ? 0 : i; }
};

// SWAP = 1 (MODE 0, Pout % 4 == 0, fp32 outputs only): the MFMA operand roles are
// exchanged -- activations feed the A side, weights the B side -- so an
// accumulator tile is [position][channel] and a lane's four consecutive registers
// are four CONSECUTIVE POSITIONS of one channel: 16 contiguous bytes of the
// (N, C, P) output, one dwordx4 store (and one dwordx4 residual load) instead of
// four dword ones.  Same products, same k order: bit-identical to SWAP = 0.
// Measured with in-kernel clock stamps on the head tower (256 x 192 tiles): the
// dword-store epilogue was 25 k of the wave's 110 k clocks (7.8 B / clk / CU:
// stores issue at ~32 clocks per wave instruction whatever their width).
template <int BM, int BN, int WM, int WN, int MODE, int SWAP>
__global__ __launch_bounds__(512) void conv_t256_c8_kernel(ConvK a) {
  static_assert(WM * WN == 8, "eight waves");
  static_assert(SWAP == 0 || MODE == 0, "swapped roles: plain convolution only");
  static_assert(BM % 64 == 0 && BN % 64 == 0, "64-row DMA chunks");
  constexpr int KB = 8;  // 16-byte k groups per 64-deep step
  constexpr int WTM = BM / WM, WTN = BN / WN;
  static_assert(WTM % 32 == 0 && WTN % 32 == 0, "wave tile");
  constexpr int TM = WTM / 32, TN = WTN / 32;
  constexpr int STAGE = KB * (BM + BN);  // 16-byte units
  typedef LoadMap<BM / 64> MA;
  typedef LoadMap<BN / 64> MB;
  __shared__ __attribute__((aligned(16))) uintx4 lds[2 * STAGE];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int l31 = lane & 31, lk = lane >> 5;
  LD_STAMP(0);
  LD_STAMP_WALL(5);
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
  const int ntaps =
      __builtin_amdgcn_readfirstlane(MODE == 1 ? a.nth * a.ntw : a.KH * a.KW);

  // ---- the positions (B rows) and weight rows (A rows) this lane addresses ---
  int bHin[MB::NPOS], bWin[MB::NPOS], boff[MB::NPOS], bh0[MB::NPOS], bw0[MB::NPOS];
#pragma unroll
  for (int q = 0; q < MB::NPOS; ++q) {
    const int ch = MB::fixed ? MB::chunk(wave, 0) : q;
    const int jb = n0 + ch * 64 + lane;
    bHin[q] = bWin[q] = boff[q] = bh0[q] = bw0[q] = 0;  // Hin = 0: never in range
    if (jb < a.J) {
      const int n = jb / a.Pout, p = jb - n * a.Pout;
      int bl, bho, bwo;
      locate_out(a.g, p, bl, bho, bwo);
      bHin[q] = a.g.lv[bl].Hin;
      bWin[q] = a.g.lv[bl].Win;
      boff[q] = n * (Cin >> 3) * Pin + a.g.lv[bl].off_in;
      if (MODE == 1) {
        bh0[q] = bho + a.ch0;
        bw0[q] = bwo + a.cw0;
      } else {
        bh0[q] = bho * a.g.stride - a.g.pad;
        bw0[q] = bwo * a.g.stride - a.g.pad;
      }
    }
  }
  unsigned va[MA::NPOS];
#pragma unroll
  for (int q = 0; q < MA::NPOS; ++q) {
    const int ch = MA::fixed ? MA::chunk(wave, 0) : q;
    const int co = m0 + ch * 64 + lane;
    va[q] = co < Cout ? (unsigned)co * 16u : kOOB;
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

  const int csteps = Cin >> 6;  // host guarantees Cin % 64 == 0
  const int nsteps = ntaps * csteps;
  int c_step = 0, c_kh = 0, c_kw = 0, c_ci0 = 0;
  // The DMA of one k-step = NT wave instructions (A rows first, then B rows),
  // issued in three groups: a burst of all of them right behind the barrier (8
  // waves x 7) queues on the CU's address unit and stalls the issuing waves in
  // front of their MFMAs (clock stamps: the main loop of the head tower is 23 %
  // shorter with the DMA removed) -- so only group 0 goes there, groups 1 and 2
  // follow under the next step's first two sub-steps.
  constexpr int NT = MA::NI + MB::NI;
  constexpr int G1 = LD_T256_SPREAD ? (NT + 2) / 3 : NT;
  constexpr int G2 = LD_T256_SPREAD ? G1 + (NT - G1 + 1) / 2 : NT;
  bool x_live = false, x_aoob = false;
  unsigned x_sa0 = 0, x_sob = 0, x_vb[MB::NPOS];
  // addresses of the next k-step (kept until its last group is issued)
  auto dma_begin = [&]() {
    x_live = c_step < nsteps;  // wave-uniform: nothing to fetch past the last step
#if defined(LD_T256_ABL) && (LD_T256_ABL & 1)  // ablation: no DMA after the first two steps
    x_live = x_live && c_step < 2;
#endif
#if defined(LD_T256_ABL) && (LD_T256_ABL & 2)  // ablation: out-of-range DMA after two steps
    const bool x_oob = c_step >= 2;
#else
    const bool x_oob = false;
#endif
    x_aoob = x_oob;
    const int wtap = MODE == 1 ? (a.kh0 + 2 * c_kh) * KW + a.kw0 + 2 * c_kw
                               : c_kh * ntw + c_kw;
    x_sa0 = (unsigned)((wtap * Kp8 + (c_ci0 >> 3)) * Cout) * 16u;
    x_sob = (unsigned)(c_ci0 >> 3) * (unsigned)Pin * 16u;
#pragma unroll
    for (int q = 0; q < MB::NPOS; ++q) {
      const int hi = bh0[q] + c_kh, wi = bw0[q] + c_kw;
      const bool ok = x_live & !x_oob & ((unsigned)hi < (unsigned)bHin[q]) &
                      ((unsigned)wi < (unsigned)bWin[q]);
      x_vb[q] = ok ? (unsigned)(boff[q] + hi * bWin[q] + wi) * 16u : kOOB;
    }
    ++c_step;
    c_ci0 += 64;
    const bool wc = c_ci0 >= Cin;
    c_ci0 = wc ? 0 : c_ci0;
    c_kw += wc ? 1 : 0;
    const bool wk = c_kw >= ntw;
    c_kw = wk ? 0 : c_kw;
    c_kh += wk ? 1 : 0;
  };
  // (branch-free: past the last step the offsets are out of range and the DMA
  // writes zeros into a stage nobody reads any more -- a branch here would cut the
  // step into basic blocks and break the MFMA / LDS interleave below)
  auto dma_ops = [&](int stage, int lo, int hi) {
    uintx4* As = lds + stage * STAGE;
    uintx4* Bs = As + KB * BM;
    const unsigned prow = (unsigned)Pin * 16u;
#pragma unroll
    for (int o = 0; o < NT; ++o) {
      if (o < lo || o >= hi) continue;
      if (o < MA::NI) {
        const int kg = MA::kg(wave, o), ch = MA::chunk(wave, o);
        dma16(rw, As + kg * BM + ch * 64, x_live && !x_aoob ? va[MA::pos(o)] : kOOB,
              x_sa0 + (unsigned)(kg * Cout) * 16u);
      } else {
        const int i = o - MA::NI;
        const int kg = MB::kg(wave, i), ch = MB::chunk(wave, i);
        dma16(rx, Bs + kg * BN + ch * 64, x_vb[MB::pos(i)], x_sob + (unsigned)kg * prow);
      }
    }
  };
  // fragments of one 16-deep sub-step: TM weight-row tiles, TN position tiles
  auto frags = [&](int stage, int s, uintx4* af, uintx4* bf) {
    const uintx4* ap = lds + stage * STAGE + wm * WTM + l31 + lk * BM;
    const uintx4* bq = lds + stage * STAGE + KB * BM + wn * WTN + l31 + lk * BN;
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = ap[2 * s * BM + i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = bq[2 * s * BN + j * 32];
  };
  auto mfmas = [&](const uintx4* af, const uintx4* bf) {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                               __builtin_bit_cast(bf16x8, bf[j]),
                               __builtin_bit_cast(bf16x8, af[i]), acc[i][j], 0, 0, 0)
                         : __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                               __builtin_bit_cast(bf16x8, af[i]),
                               __builtin_bit_cast(bf16x8, bf[j]), acc[i][j], 0, 0, 0);
  };
  // one fragment read behind each of the first TM + TN MFMAs of a sub-step
  auto interleave = [&]() {
#pragma unroll
    for (int i = 0; i < TM + TN; ++i) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // MFMA
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
    }
    __builtin_amdgcn_sched_group_barrier(0x008, TM * TN - TM - TN, 0);
  };
  // Step u: three sub-steps, each multiplying the fragments read during the
  // previous one; then the ONE barrier of the step -- by then this wave's reads of
  // stage S are complete (lgkmcnt) and its share of step u + 1 has landed (vmcnt);
  // the fourth sub-step runs over the reads of step u + 1's first fragments and
  // the DMA issue of step u + 2 into the stage everybody just left.
  uintx4 afA[TM], bfA[TN], afB[TM], bfB[TN];
#ifdef LD_T256_STAMP
  long long w_lgkm = 0, w_vm = 0, w_bar = 0;
#endif
  auto step = [&](int S) {
    frags(S, 1, afB, bfB);
    dma_ops(S ^ 1, G1, G2);  // step u + 1, second group
    mfmas(afA, bfA);
    interleave();
    frags(S, 2, afA, bfA);
    dma_ops(S ^ 1, G2, NT);  // step u + 1, third group
    mfmas(afB, bfB);
    interleave();
    frags(S, 3, afB, bfB);
    mfmas(afA, bfA);
    interleave();
#ifdef LD_T256_STAMP
    {
      const long long t0 = clock64();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long long t1 = clock64();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long t2 = clock64();
      asm volatile("s_barrier" ::: "memory");
      const long long t3 = clock64();
      w_lgkm += t1 - t0;
      w_vm += t2 - t1;
      w_bar += t3 - t2;
    }
#else
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
    frags(S ^ 1, 0, afA, bfA);
    dma_begin();
    dma_ops(S, 0, G1);  // step u + 2, first group
    mfmas(afB, bfB);
  };

  LD_STAMP(1);
  dma_begin();
  dma_ops(0, 0, NT);
  dma_begin();
  dma_ops(1, 0, G1);
  // the first step's share has landed once at most the second's is outstanding
  if (nsteps > 1)
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G1) : "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_barrier" ::: "memory");
  LD_STAMP(2);
  frags(0, 0, afA, bfA);
  for (int u = 0;;) {
    step(0);
    if (++u >= nsteps) break;
    step(1);
    if (++u >= nsteps) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LD_STAMP(3);
#ifdef LD_T256_STAMP
  LD_STAMP_VAL(8, w_lgkm);
  LD_STAMP_VAL(9, w_vm);
  LD_STAMP_VAL(10, w_bar);
#endif

  const rsrc_t r_sc = make_rsrc(a.scale, a.scale ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_sh = make_rsrc(a.shift, a.scale ? (unsigned)Cout * 4u : 0u);
  const rsrc_t r_bi = make_rsrc(a.bias, a.bias ? (unsigned)Cout * 4u : 0u);
  if constexpr (SWAP) {
    // ---- epilogue, swapped roles.  In registers a lane holds one channel x four
    // consecutive positions (x 4 groups x TM x TN tiles); stored like that, a wave
    // instruction would touch 64 different cache lines with 16 bytes each (measured:
    // slower than dword stores on the 256 x 192 tile).  So each 32 x 32 tile takes a
    // turn through a wave-private LDS buffer [channel][32 positions + 4 pad]: four
    // 16-byte writes, four 16-byte reads that hand every lane (channel = lane / 8,
    // positions 4 (lane % 8) ...) -- a wave instruction then stores eight whole
    // 128-byte rows.  No barrier: the buffers are private to the wave; the one
    // below separates them from the main loop's last fragment reads.
    __builtin_amdgcn_s_barrier();
    const bool has_res = a.residual != nullptr;
    const bool has_y = a.y != nullptr;
    const bool has_raw = a.y_raw != nullptr;
    const bool relu = a.relu != 0;
    const bool has_aff = a.scale != nullptr;
    const int Pout = __builtin_amdgcn_readfirstlane(a.Pout);
    const float rcpP = 1.0f / (float)Pout;
    constexpr int TROW = 36;                 // floats per channel row (32 + pad)
    constexpr int TBUF = 32 * TROW;          // floats per tile buffer
    static_assert(8 * TN * TBUF * 4 <= 2 * STAGE * 16, "epilogue staging fits the LDS");
    float* tb = (float*)lds + wave * (TN * TBUF);
    const int q8 = lane & 7, c8 = lane >> 3;
    size_t off[TN];  // element offset of (n, channel 0, p) of this lane's 4 positions
    bool ok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int jc = n0 + wn * WTN + j * 32 + 4 * q8;
      ok[j] = jc < a.J;  // J % 4 == 0: a group of four is valid or absent as a whole
      int n = (int)((float)jc * rcpP);
      int pp = jc - n * Pout;
      if (pp < 0) { --n; pp += Pout; }
      if (pp >= Pout) { ++n; pp -= Pout; }
      off[j] = (size_t)n * Cout * Pout + pp;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      // registers -> LDS: channel l31, positions 8 g + 4 lk + (0 .. 3)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          floatx4_t v4;
#pragma unroll
          for (int e = 0; e < 4; ++e) v4[e] = acc[i][j][4 * g + e];
          *(floatx4_t*)(tb + j * TBUF + l31 * TROW + 8 * g + 4 * lk) = v4;
        }
      float sc[4], sh[4];
      size_t crow[4];
      bool cok[4];
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int c = m0 + wm * WTM + i * 32 + rr * 8 + c8;
        cok[rr] = c < Cout;
        const unsigned ro = (unsigned)c * 4u;  // >= Cout / absent operand: zeros
        sc[rr] = has_aff ? buf_load(r_sc, ro, 0) : 1.0f;
        sh[rr] = buf_load(r_sh, ro, 0) + buf_load(r_bi, ro, 0);
        crow[rr] = (size_t)c * Pout;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        floatx4_t rv[4];
        __builtin_amdgcn_sched_barrier(0);
        if (has_res) {
#pragma unroll
          for (int rr = 0; rr < 4; ++rr)
            rv[rr] = (cok[rr] && ok[j]) ? *(const floatx4_t*)(a.residual + off[j] + crow[rr])
                                        : floatx4_t{0.0f, 0.0f, 0.0f, 0.0f};
        }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const floatx4_t raw =
              *(const floatx4_t*)(tb + j * TBUF + (rr * 8 + c8) * TROW + 4 * q8);
          if (!(cok[rr] && ok[j])) continue;
          floatx4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = raw[e] * sc[rr] + sh[rr];
            if (has_res) v[e] += rv[rr][e];
            if (relu) v[e] = fmaxf(v[e], 0.0f);
          }
          if (has_raw) *(floatx4_t*)(a.y_raw + off[j] + crow[rr]) = raw;
          if (has_y) *(floatx4_t*)(a.y + off[j] + crow[rr]) = v;
        }
      }
    }
#ifdef LD_T256_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    LD_STAMP(4);
    LD_STAMP_WALL(6);
#endif
    return;
  }
  // ---- epilogue: direct stores, 32 consecutive positions per accumulator row
  // (the loop order and the branch-free operand loads of conv_tile_c8_kernel)
  const bool has_res = a.residual != nullptr;
  const bool res8 = a.res_c8 != nullptr;
  const bool has_y = a.y != nullptr;
  const bool c8out = a.y_c8 != nullptr;
  const bool relu = a.relu != 0;
  const bool has_aff = a.scale != nullptr, has_bias = a.bias != nullptr;
  size_t colbase[TN], c8base[TN];
  bool jok[TN];
  const int prow = MODE == 1 ? a.Pfull : a.Pout;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int jc = n0 + wn * WTN + j * 32 + l31;
    jok[j] = jc < a.J;
    const int n = jc / a.Pout;
    int p = jc - n * a.Pout;
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
    const int rbase = m0 + wm * WTM + i * 32 + 4 * lk;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row0 = rbase + 8 * g;
      const size_t c8row = (size_t)(row0 >> 3) * a.Pout * 16 + (row0 & 4) * 2;
      float sc[4], sh[4], bi[4], rv[TN][4];
      uintx2 rraw[TN];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const unsigned ro = (unsigned)(row0 + e) * 4u;
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
        if (MODE == 0 && c8out && row0 < Cout)
          *(uintx2*)((char*)a.y_c8 + c8base[j] + c8row) =
              __builtin_bit_cast(uintx2, __builtin_convertvector(q, bf16x4));
        if (MODE == 0 && a.raw_c8 && row0 < Cout) {  // as in conv_tile_c8_kernel
          const floatx4_t rw = {acc[i][j][4 * g], acc[i][j][4 * g + 1],
                                acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
          *(uintx2*)((char*)a.raw_c8 + c8base[j] + c8row) =
              __builtin_bit_cast(uintx2, __builtin_convertvector(rw, bf16x4));
        }
      }
    }
  }
#ifdef LD_T256_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  LD_STAMP(4);
  LD_STAMP_WALL(6);
#endif
}

}  // namespace

#ifdef LD_T256_STAMP
extern "C" int ld_debug_t256_stamps(void* p) {
  return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_t256_stamps), &p, sizeof(p));
}
#endif

// (BM, BN, WM, WN): workgroup tile and the wave grid inside it
#define LD_T256_SHAPES(X) \
  X(256, 256, 2, 4) X(256, 192, 4, 2) X(128, 256, 2, 4) X(256, 128, 4, 2)

bool ld_bf16_t256_fits(const ConvK& k, int BM, int BN) {
  if (!k.x_c8 || k.Cin % 64 != 0) return false;
  const int cout32 = (k.Cout + 31) / 32 * 32;
  if (BM > cout32 + 32) return false;
#define LD_CASE(BM_, BN_, WM_, WN_) \
  if (BM == BM_ && BN == BN_) return true;
  LD_T256_SHAPES(LD_CASE)
#undef LD_CASE
  return false;
}

int ld_bf16_t256_launch(int mode, const ConvK& k, int BM, int BN, hipStream_t stream) {
  if (!ld_bf16_t256_fits(k, BM, BN)) return LD_EUNSUPPORTED;
  const int nb = ((k.Cout + BM - 1) / BM) * ((k.J + BN - 1) / BN);
  // swapped operand roles (16-byte stores of the fp32 output) wherever the output
  // rows allow it; LD_CONV_T256_SWAP=0 turns it off (A/B runs)
  static const bool allow_swap = [] {
    const char* e = getenv("LD_CONV_T256_SWAP");
    return !(e && e[0] == '0');
  }();
  const bool swap = allow_swap && mode == 0 && k.Pout % 4 == 0 && !k.y_c8 && !k.res_c8 &&
                    !k.raw_c8;
#define LD_CASE(BM_, BN_, WM_, WN_)                                                \
  if (BM == BM_ && BN == BN_) {                                                    \
    if (mode == 1)                                                                 \
      LD_LAUNCH((conv_t256_c8_kernel<BM_, BN_, WM_, WN_, 1, 0>), dim3(nb),         \
                dim3(512), 0, stream, k);                                          \
    else if (swap)                                                                 \
      LD_LAUNCH((conv_t256_c8_kernel<BM_, BN_, WM_, WN_, 0, 1>), dim3(nb),         \
                dim3(512), 0, stream, k);                                          \
    else                                                                           \
      LD_LAUNCH((conv_t256_c8_kernel<BM_, BN_, WM_, WN_, 0, 0>), dim3(nb),         \
                dim3(512), 0, stream, k);                                          \
    return (int)hipGetLastError();                                                 \
  }
  LD_T256_SHAPES(LD_CASE)
#undef LD_CASE
  return LD_EUNSUPPORTED;
}
