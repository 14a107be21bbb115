// fp32 weight gradient of every trainable conv on the LD train step, workgroup
// tiled with an in-workgroup split of the reduction (round 4).
//
// Replaces the weight half of nn.Conv2d's backward under
//   mmdet/models/backbones/resnet.py:260-299 (Bottleneck), necks/fpn.py:170-221,
//   dense_heads/gfl_head.py:102-133 (towers / predictors).
//
//   dW[co][ci][tap] = sum_j dY[co][j] * X[ci][pos(j, tap)],   j = n * Pout + p
//
// Both operands are contiguous along the REDUCTION index j (NCHW), the f32 MFMA
// wants a lane per ROW: the tiles go through LDS ([row][j] image, coalesced
// 128-byte row segments in, fragment reads [row = lane][k] out).  Round 1-3 gave
// every wavefront a private 64 x 64 tile (conv_wgrad_wave_kernel, conv.hip): 64
// global loads + 64 LDS writes per 64 MFMAs, 174 VGPRs = two waves per SIMD, one
// 16 KB partial slab per WAVE summed by a second launch (2.4 ms of the 8.5 ms
// weight-gradient time of a C2 step).  Here:
//   * a workgroup of 4 * KG wavefronts owns a 128(co) x 128(ci) tile of one tap;
//     a staged dword feeds two waves (half the loads and LDS writes per MFMA);
//   * KG k-groups of 4 waves split every staged j-slice between them (group g
//     reads the fragments of j in [g * BK / KG, (g + 1) * BK / KG)) and their
//     accumulators are summed through LDS in fixed order at the end: one 64 KB
//     partial per 4 * KG waves instead of one 16 KB slab per wave -- the
//     cross-workgroup split count, and with it the partial traffic, drops by KG;
//   * 16 staging registers per lane (KG = 2) instead of 64: <= 128 VGPRs, four
//     waves per SIMD; the loads of slice u + 2 are issued while slice u is
//     multiplied and slice u + 1 is written to the other LDS buffer (one barrier
//     per slice), so a load has a whole slice of every resident wave to land;
//   * fragment reads are ds_read_b64 (two consecutive k per lane feed two
//     MFMAs: lanes < 32 take k = 4q, 4q + 1, lanes >= 32 take 4q + 2, 4q + 3;
//     the pairing is the same on both operands, which is all the MFMA needs);
//     row pitch BK + 2 floats: the 32 rows of a fragment cover all 64 banks;
//   * cross-workgroup partials are combined either by the fixed-order slab
//     reduce launch (out_mode 0), written straight to dW when the layer needs no
//     split (1), or inside this launch by the LAST-ARRIVING workgroup of a tile
//     (2): partial -> agent-scope release -> ticket; the workgroup that draws the
//     last ticket sums all partials in split order (its own included, from
//     memory) and writes dW.  No float atomics, no spinning: bit-reproducible
//     and placement-independent (cdna_hip_programming.md Guideline 16).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "conv_common.h"

namespace {

typedef float floatx2 __attribute__((ext_vector_type(2)));

struct WgradTileOut {
  float* dw;          // (Cout, Cin, KH, KW)
  float* part;        // out_mode 2: [split][tile][wave][16][64 lanes][4] partials
  unsigned* tickets;  // out_mode 2: one counter per tile, zero on entry, left zero
  int accumulate;
  int out_mode;       // 0 slabs + reduce launch, 1 direct (splits == 1), 2 fused
};

// waves per SIMD the register allocation must leave room for: LDS admits two
// workgroups per CU at BK = 32 and one at BK = 64; the 64 accumulators cap it at 4
constexpr int wgrad_tile_occ(int kg, int bk) {
  const int wgs = bk == 32 ? 2 : 1;
  return wgs * kg > 4 ? 4 : wgs * kg;
}

// DBG (timing attribution only, results are wrong when non-zero; LD_WGRAD_DBG in
// tools/): 1 = no barrier in the main loop, 2 = no global loads, 4 = no LDS
// writes, 8 = no per-slice position decode, 16 = no fragment reads.
template <int KG, int BK, bool ML, int DBG = 0>
__global__ __launch_bounds__(256 * KG, wgrad_tile_occ(KG, BK)) void conv_wgrad_tile_kernel(
    WgradK a, WgradTileOut o) {
  constexpr int TB = 128;            // tile edge (co and ci)
  constexpr int NT = 256 * KG;       // threads
  constexpr int LDA = BK + 2;        // LDS row pitch in floats (8-byte aligned rows)
  constexpr int RPASS = NT / BK;     // rows one load pass of the workgroup covers
  constexpr int NP = TB / RPASS;     // passes per operand and slice
  constexpr int OPF = TB * LDA;      // floats per operand image
  constexpr int STAGE = 2 * OPF;     // dY image, then X image
  constexpr int KPG = BK / KG;       // j per k-group and slice
  constexpr int NQ = KPG / 4;        // fragment quads (4 k = 8 MFMAs) per slice
  static_assert(NP * 2 == NQ * 4, "four staging slots per quad");
  static_assert(NQ >= 2 && NQ % 2 == 0, "fragment double buffer");
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
  __shared__ int s_geo[LD_MAX_LEVELS * 6];
  __shared__ unsigned s_last;

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int kg = wave >> 2, w4 = wave & 3;
  const int wm = w4 >> 1, wn = w4 & 1;
  const int l31 = lane & 31, lk = lane >> 5;
  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Pout = __builtin_amdgcn_readfirstlane(a.Pout);
  const int nlev = __builtin_amdgcn_readfirstlane(a.g.num_levels);
  const int stride = __builtin_amdgcn_readfirstlane(a.g.stride);
  const int pad = __builtin_amdgcn_readfirstlane(a.g.pad);
  const int mt = (Cout + TB - 1) / TB, nt = (Cin + TB - 1) / TB;
  const int ntaps = a.KH * a.KW;
  const int ntiles = mt * nt * ntaps;
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_id = b % ntiles;
  const int split = b / ntiles;
  const int ntile = tile_id % nt;
  const int mtile = (tile_id / nt) % mt;
  const int tap = tile_id / (nt * mt);
  const int m0 = mtile * TB, c0 = ntile * TB;
  const int kh = tap / a.KW, kw = tap - kh * a.KW;
  const int jbeg = split * a.jchunk;
  const int jend = min(a.J, jbeg + a.jchunk);

  if (ML) {
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
  }
  const int Hin0 = __builtin_amdgcn_readfirstlane(a.g.lv[0].Hin);
  const int Win0 = __builtin_amdgcn_readfirstlane(a.g.lv[0].Win);
  const int Wout0 = __builtin_amdgcn_readfirstlane(a.g.lv[0].Wout);

  // ---- load side: lane = (j offset kq of the slice, first row r0) ---------
  const int kq = t % BK;
  const int r0 = t / BK;
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t ry = make_rsrc(a.dy, a.dy_bytes);
  // Rows past the channel count (ragged last tile) need no mask: the row offset
  // travels in the range-checked voffset, so such a row reads either the next
  // image's rows (finite garbage that only reaches accumulator rows / columns
  // >= Cout / Cin, which are never stored) or, past the tensor's end, the
  // descriptor's zero.  (An soffset is NOT range-checked: it would read beyond
  // the allocation.)
  const unsigned da = (unsigned)RPASS * (unsigned)Pout * 4u;
  const unsigned db = (unsigned)RPASS * (unsigned)Pin * 4u;

  // byte offsets of (dY row m0 + r0, X row c0 + r0) at column j0 + kq; kOOB (the
  // descriptor's out-of-range zero) past the split's end and in the padding
  auto decode = [&](int j0, unsigned& vy, unsigned& vx) {
    const int j = j0 + kq;
    const int n = j / Pout, p = j - n * Pout;
    int Hin = Hin0, Win = Win0, Wout = Wout0, off_in = 0, off_out = 0;
    if (ML) {
      int l = 0;
      for (int i = 1; i < nlev; ++i)
        if (p >= s_geo[i * 6 + 5]) l = i;
      Hin = s_geo[l * 6 + 0];
      Win = s_geo[l * 6 + 1];
      Wout = s_geo[l * 6 + 3];
      off_in = s_geo[l * 6 + 4];
      off_out = s_geo[l * 6 + 5];
    }
    const int r = p - off_out;
    const int ho = r / Wout, wo = r - ho * Wout;
    const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
    const bool jok = j < jend;
    const bool xok = jok && hi >= 0 && hi < Hin && wi >= 0 && wi < Win;
    vy = jok ? (unsigned)((n * Cout + m0 + r0) * Pout + p) * 4u : kOOB;
    vx = xok ? (unsigned)((n * Cin + c0 + r0) * Pin + off_in + hi * Win + wi) * 4u
             : kOOB;
  };

  float sta[NP], stb[NP];  // the slice in flight: dY rows, X rows
  float sta2[NP], stb2[NP];            // DBG & 64 only
#pragma unroll
  for (int i = 0; i < NP; ++i) sta2[i] = stb2[i] = 0.0f;
  floatx4 sta4[NP / 4], stb4[NP / 4];  // DBG & 32 only
#pragma unroll
  for (int i = 0; i < NP / 4; ++i) sta4[i] = stb4[i] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
  unsigned vy = kOOB, vx = kOOB;
  // LDS write position of pass 0 (floats): image row r0, column kq
  const int wr0 = r0 * LDA + kq;
  // fragment read positions (floats) of this lane: row l31 of MFMA tile 0 of the
  // wave's half, columns of k-group kg, the lane half's two k
  const int fa0 = (wm * 64 + l31) * LDA + kg * KPG + 2 * lk;
  const int fb0 = OPF + (wn * 64 + l31) * LDA + kg * KPG + 2 * lk;

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nsteps = (jend - jbeg + BK - 1) / BK;

  // ---- prologue: slice 0 -> LDS buffer 0, slice 1 -> registers --------------
  {
    decode(jbeg, vy, vx);
    unsigned sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      sta[i] = buf_load(ry, vy + sa, 0);
      stb[i] = buf_load(rx, vx + sb, 0);
      sa += da;
      sb += db;
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      lds[wr0 + i * RPASS * LDA] = sta[i];
      lds[OPF + wr0 + i * RPASS * LDA] = stb[i];
    }
    decode(jbeg + BK, vy, vx);
    sa = 0;
    sb = 0;
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      sta[i] = buf_load(ry, vy + sa, 0);
      stb[i] = buf_load(rx, vx + sb, 0);
      sa += da;
      sb += db;
    }
  }
  __syncthreads();

  // ---- main loop: slice `step` from LDS buffer step & 1 ---------------------
  // Per fragment quad: 4 ds_read_b64 for the next quad, two (LDS write of slice
  // step + 1, re-issue for slice step + 2) slots per operand, 8 MFMAs.
  for (int base = 0; base < nsteps; base += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int step = base + u;
      if (step >= nsteps) break;
      const float* cur = lds + u * STAGE;
      float* nxt = lds + (u ^ 1) * STAGE;
      unsigned vy2 = vy, vx2 = vx;
      if (!(DBG & 8)) decode(jbeg + (step + 2) * BK, vy2, vx2);
      floatx2 fa[2][2], fb[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[0][i] = *reinterpret_cast<const floatx2*>(cur + fa0 + i * 32 * LDA);
        fb[0][i] = *reinterpret_cast<const floatx2*>(cur + fb0 + i * 32 * LDA);
      }
      unsigned sa = 0, sb = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c = q & 1;
        if (q + 1 < NQ && !(DBG & 16)) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            fa[c ^ 1][i] = *reinterpret_cast<const floatx2*>(cur + fa0 + i * 32 * LDA +
                                                            4 * (q + 1));
            fb[c ^ 1][i] = *reinterpret_cast<const floatx2*>(cur + fb0 + i * 32 * LDA +
                                                            4 * (q + 1));
          }
        }
        if constexpr ((DBG & 32) != 0) {
          // timing experiment: the same bytes as 16-byte loads (8 rows x 128 B per
          // wave instruction), taps' border masks ignored
          constexpr int NP4 = NP / 4;
          const int row4 = t / 8, quad = t % 8;
          const bool isb = q >= NP4;
          const int i4 = isb ? q - NP4 : q;
          float* dst = nxt + (isb ? OPF : 0) + (row4 + i4 * (NT / 8)) * LDA + 4 * quad;
          floatx4& reg = isb ? stb4[i4] : sta4[i4];
          if (!(DBG & 4)) {
#pragma unroll
            for (int e = 0; e < 4; ++e) dst[e] = reg[e];
          }
          const unsigned vb = isb ? vx2 - (unsigned)(r0 * Pin + kq) * 4u +
                                        (unsigned)(row4 * Pin + 4 * quad) * 4u
                                  : vy2 - (unsigned)(r0 * Pout + kq) * 4u +
                                        (unsigned)(row4 * Pout + 4 * quad) * 4u;
          const unsigned so = (unsigned)i4 * (unsigned)(NT / 8) * (unsigned)(isb ? Pin : Pout) * 4u;
          if (!(DBG & 2))
            reg = __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                                                  isb ? rx : ry, vb + so, 0, 0));
        } else
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = 2 * q + h;
          if (!(DBG & 4)) {
            nxt[wr0 + i * RPASS * LDA] = sta[i];
            nxt[OPF + wr0 + i * RPASS * LDA] = stb[i];
          }
          if constexpr ((DBG & 64) != 0) {  // timing: one more slice of load distance
            sta[i] = sta2[i];
            stb[i] = stb2[i];
            sta2[i] = buf_load(ry, vy2 + sa, 0);
            stb2[i] = buf_load(rx, vx2 + sb, 0);
          } else if (!(DBG & 2)) {
            sta[i] = buf_load(ry, vy2 + sa, 0);
            stb[i] = buf_load(rx, (DBG & 128) ? kOOB : vx2 + sb, 0);
          }
          sa += da;
          sb += db;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][e], fb[c][j][e],
                                                               acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!(DBG & 1)) __syncthreads();  // slice step + 1 complete; nobody still reads buffer u
    }
  }

  // ---- k-groups -> one accumulator set (fixed order g0 + g1 + g2 + ...) ------
  // the operand images are dead (the loop's last barrier): 64 KB of them hold
  // one group's accumulators, [w4][register][lane]
  if (KG > 1) {
    float* red = lds;
#pragma unroll 1
    for (int g = 1; g < KG; ++g) {
      if (kg == g) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              red[(w4 * 64 + (i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
      }
      __syncthreads();
      if (kg == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              acc[i][j][r] += red[(w4 * 64 + (i * 2 + j) * 16 + r) * 64 + lane];
      }
      __syncthreads();
    }
  }

  // ---- output ---------------------------------------------------------------
  // accumulator (i, j, r) of lane (l31, lk): co = m0 + wm*64 + i*32 + (r & 3) +
  // 8 (r >> 2) + 4 lk, ci = c0 + wn*64 + j*32 + l31
  auto store_dw = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = c0 + wn * 64 + j * 32 + l31;
      if (ci >= Cin) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (co >= Cout) continue;
          const size_t idx = ((size_t)co * Cin + ci) * ntaps + tap;
          o.dw[idx] = o.accumulate ? o.dw[idx] + acc[i][j][r] : acc[i][j][r];
        }
    }
  };
  if (o.out_mode == 0) {
    if (kg == 0) {
      // slab [split][tap][co][ci], ci fastest (= lane & 31)
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
    return;
  }
  if (o.out_mode == 1) {
    if (kg == 0) store_dw();
    return;
  }
  // fused: publish the partial in register layout (16-byte stores, 1 KB per wave
  // instruction), release, draw a ticket
  const int nsplit = a.splits;
  floatx4* part = reinterpret_cast<floatx4*>(o.part);
  if (kg == 0) {
    floatx4* mine = part + (((size_t)split * ntiles + tile_id) * 4 + w4) * 16 * 64 + lane;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          floatx4 v = {acc[i][j][4 * r4], acc[i][j][4 * r4 + 1], acc[i][j][4 * r4 + 2],
                       acc[i][j][4 * r4 + 3]};
          mine[((i * 2 + j) * 4 + r4) * 64] = v;
        }
  }
  __syncthreads();
  if (t == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned old = __hip_atomic_fetch_add(o.tickets + tile_id, 1u, __ATOMIC_RELAXED,
                                                __HIP_MEMORY_SCOPE_AGENT);
    s_last = (old == (unsigned)(nsplit - 1)) ? 1u : 0u;
  }
  __syncthreads();
  if (s_last == 0u) return;
  if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  __syncthreads();
  // Every wave of the workgroup sums its share of the tile: chunk c = (i, j, r4)
  // of wave position w4 goes to k-group c % KG.  Splits are added in index order
  // (the own partial too, from memory): the result does not depend on which
  // workgroup arrived last.  Partials come through a buffer descriptor (lane
  // offset in the VGPR, everything else scalar); CH chunks x two splits in
  // flight per lane.
  {
    constexpr int NC = 16 / KG;          // chunks of this wave
    constexpr int CH = KG == 1 ? 8 : 4;  // chunks per pass (register budget)
    const unsigned sp_bytes = (unsigned)ntiles * 4u * 16u * 1024u;  // bytes per split
    const rsrc_t rp = make_rsrc(o.part, sp_bytes * (unsigned)nsplit);
    const unsigned voff = (unsigned)lane * 16u;
    const unsigned base = ((unsigned)tile_id * 4u + (unsigned)w4) * 16u * 1024u;
#pragma unroll 1
    for (int c0c = 0; c0c < NC; c0c += CH) {
      floatx4 sum[CH], va[CH], vb[CH];
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        sum[cc] = floatx4{0.0f, 0.0f, 0.0f, 0.0f};
        va[cc] = __builtin_bit_cast(
            floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                         rp, voff, base + (unsigned)((c0c + cc) * KG + kg) * 1024u, 0));
      }
#pragma unroll 1
      for (int sp = 0; sp < nsplit; sp += 2) {
        // a split index past the end reads the descriptor's out-of-range zero
        const unsigned s1 = base + (unsigned)(sp + 1) * sp_bytes;
        const unsigned s2 = base + (unsigned)(sp + 2) * sp_bytes;
        const unsigned v1 = sp + 1 < nsplit ? voff : kOOB;
        const unsigned v2 = sp + 2 < nsplit ? voff : kOOB;
#pragma unroll
        for (int cc = 0; cc < CH; ++cc)
          vb[cc] = __builtin_bit_cast(
              floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                           rp, v1, s1 + (unsigned)((c0c + cc) * KG + kg) * 1024u, 0));
#pragma unroll
        for (int cc = 0; cc < CH; ++cc) sum[cc] += va[cc];
#pragma unroll
        for (int cc = 0; cc < CH; ++cc)
          va[cc] = __builtin_bit_cast(
              floatx4, __builtin_amdgcn_raw_buffer_load_b128(
                           rp, v2, s2 + (unsigned)((c0c + cc) * KG + kg) * 1024u, 0));
        if (sp + 1 < nsplit) {
#pragma unroll
          for (int cc = 0; cc < CH; ++cc) sum[cc] += vb[cc];
        }
      }
#pragma unroll
      for (int cc = 0; cc < CH; ++cc) {
        const int c = (c0c + cc) * KG + kg;
        const int i = c >> 3, j = (c >> 2) & 1, r4 = c & 3;
        const int ci = c0 + wn * 64 + j * 32 + l31;
        if (ci >= Cin) continue;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int co = m0 + wm * 64 + i * 32 + e + 8 * r4 + 4 * lk;
          if (co >= Cout) continue;
          const size_t idx = ((size_t)co * Cin + ci) * ntaps + tap;
          o.dw[idx] = o.accumulate ? o.dw[idx] + sum[cc][e] : sum[cc][e];
        }
      }
    }
  }
  if (t == 0)
    __hip_atomic_store(o.tickets + tile_id, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- 3x3 convs: the three kw taps of a kernel row in ONE workgroup ------------
// What the timing attribution of the kernel above shows (head-tower shape,
// profiles/r04_wgrad_attribution.txt): MFMAs alone 369 us, everything but the
// global loads 376 us, the full kernel 460 us; with the X loads returning nothing
// 405 us; one more slice of load distance changes nothing -- the cost is
// proportional to the BYTES that miss L1 (8 B / clk / CU at 128 x 128 tiles),
// not to latency, barriers or LDS work.  The nine tap workgroups of a (co, ci)
// tile all read the same dY rows and -- shifted by one position -- the same X
// rows; spread over different CUs each of them misses L1 on its own.  Here the
// three kw taps of one kernel row share a workgroup of 12 wavefronts: the dY
// image is staged once for all three (a third of its L1 misses) and the three X
// images are loaded back to back by the same CU (the shifted segments are the
// same cache lines: L1 hits for two of the three).  Each 4-wave tap group owns
// its X image, its tap and its 128 x 128 accumulators; nothing is reduced across
// groups.  LDS per slice: dY 144 rows (18 passes of 8, so that the three groups
// stage six passes each -- rows 128..143 are loaded and never read) + 3 x 128
// X rows, pitch 34 floats: 71.8 KB, two buffers, one workgroup per CU, three
// waves per SIMD.
template <bool ML, int DBG = 0>
__global__ __launch_bounds__(768, 3) void conv_wgrad_tap3_kernel(WgradK a, WgradTileOut o) {
  constexpr int TB = 128, BK = 32, LDA = BK + 2;
  constexpr int DYR = 144;              // dY image rows (18 passes x 8)
  constexpr int NPX = 16, NPY = 6;      // load passes per lane: X image, dY share
  constexpr int XOFF = DYR * LDA;       // first X image (floats)
  constexpr int XIMG = TB * LDA;        // one X image
  constexpr int STAGE = XOFF + 3 * XIMG;
  constexpr int NQ = BK / 4;            // 8 fragment quads per slice
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];
  __shared__ int s_geo[LD_MAX_LEVELS * 6];

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int g = wave >> 2, w4 = wave & 3;  // tap group (= kw), wave position
  const int wm = w4 >> 1, wn = w4 & 1;
  const int l31 = lane & 31, lk = lane >> 5;
  const int Cin = __builtin_amdgcn_readfirstlane(a.Cin);
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int Pin = __builtin_amdgcn_readfirstlane(a.Pin);
  const int Pout = __builtin_amdgcn_readfirstlane(a.Pout);
  const int nlev = __builtin_amdgcn_readfirstlane(a.g.num_levels);
  const int stride = __builtin_amdgcn_readfirstlane(a.g.stride);
  const int pad = __builtin_amdgcn_readfirstlane(a.g.pad);
  const int mt = (Cout + TB - 1) / TB, nt = (Cin + TB - 1) / TB;
  const int ntiles = mt * nt * 3;
  int b = xcd_swizzle(blockIdx.x, gridDim.x);
  const int tile_id = b % ntiles;
  const int split = b / ntiles;
  const int ntile = tile_id % nt;
  const int mtile = (tile_id / nt) % mt;
  const int kh = tile_id / (nt * mt);
  const int kw = g;
  const int tap = kh * 3 + kw;
  const int m0 = mtile * TB, c0 = ntile * TB;
  const int jbeg = split * a.jchunk;
  const int jend = min(a.J, jbeg + a.jchunk);

  if (ML) {
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
  }
  const int Hin0 = __builtin_amdgcn_readfirstlane(a.g.lv[0].Hin);
  const int Win0 = __builtin_amdgcn_readfirstlane(a.g.lv[0].Win);
  const int Wout0 = __builtin_amdgcn_readfirstlane(a.g.lv[0].Wout);

  // load side: the group's 256 lanes = (column kq of the slice, first row r0)
  const int tl = t & 255;
  const int kq = tl & 31, r0 = tl >> 5;
  const int ry0 = r0 + 48 * g;  // first dY row of this lane: passes 6g .. 6g + 5
  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t ry = make_rsrc(a.dy, a.dy_bytes);
  const unsigned da = 8u * (unsigned)Pout * 4u, db = 8u * (unsigned)Pin * 4u;

  auto decode = [&](int j0, unsigned& vy, unsigned& vx) {
    const int j = j0 + kq;
    const int n = j / Pout, p = j - n * Pout;
    int Hin = Hin0, Win = Win0, Wout = Wout0, off_in = 0, off_out = 0;
    if (ML) {
      int l = 0;
      for (int i = 1; i < nlev; ++i)
        if (p >= s_geo[i * 6 + 5]) l = i;
      Hin = s_geo[l * 6 + 0];
      Win = s_geo[l * 6 + 1];
      Wout = s_geo[l * 6 + 3];
      off_in = s_geo[l * 6 + 4];
      off_out = s_geo[l * 6 + 5];
    }
    const int r = p - off_out;
    const int ho = r / Wout, wo = r - ho * Wout;
    const int hi = ho * stride - pad + kh, wi = wo * stride - pad + kw;
    const bool jok = j < jend;
    const bool xok = jok && hi >= 0 && hi < Hin && wi >= 0 && wi < Win;
    // dY rows past Cout (the image's rows 128..143, ragged tiles): range-checked
    // voffset -> the next image's rows or zero, never stored (see the kernel above)
    vy = jok ? (unsigned)((n * Cout + m0 + ry0) * Pout + p) * 4u : kOOB;
    vx = xok ? (unsigned)((n * Cin + c0 + r0) * Pin + off_in + hi * Win + wi) * 4u
             : kOOB;
  };

  float sty[NPY], stx[NPX];
  unsigned vy = kOOB, vx = kOOB;
  const int wry = ry0 * LDA + kq;                   // dY image, pass 0
  const int wrx = XOFF + g * XIMG + r0 * LDA + kq;  // this group's X image, pass 0
  const int fa0 = (wm * 64 + l31) * LDA + 2 * lk;
  const int fb0 = XOFF + g * XIMG + (wn * 64 + l31) * LDA + 2 * lk;

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int nsteps = (jend - jbeg + BK - 1) / BK;
  {
    decode(jbeg, vy, vx);
    unsigned sa = 0, sb = 0;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      stx[i] = buf_load(rx, vx + sb, 0);
      sb += db;
      if (i < NPY) {
        sty[i] = buf_load(ry, vy + sa, 0);
        sa += da;
      }
    }
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      lds[wrx + i * 8 * LDA] = stx[i];
      if (i < NPY) lds[wry + i * 8 * LDA] = sty[i];
    }
    decode(jbeg + BK, vy, vx);
    sa = 0;
    sb = 0;
#pragma unroll
    for (int i = 0; i < NPX; ++i) {
      stx[i] = buf_load(rx, vx + sb, 0);
      sb += db;
      if (i < NPY) {
        sty[i] = buf_load(ry, vy + sa, 0);
        sa += da;
      }
    }
  }
  __syncthreads();

  for (int base = 0; base < nsteps; base += 2) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int step = base + u;
      if (step >= nsteps) break;
      const float* cur = lds + u * STAGE;
      float* nxt = lds + (u ^ 1) * STAGE;
      unsigned vy2, vx2;
      decode(jbeg + (step + 2) * BK, vy2, vx2);
      floatx2 fa[2][2], fb[2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        fa[0][i] = *reinterpret_cast<const floatx2*>(cur + fa0 + i * 32 * LDA);
        fb[0][i] = *reinterpret_cast<const floatx2*>(cur + fb0 + i * 32 * LDA);
      }
      unsigned sa = 0, sb = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const int c = q & 1;
        if (q + 1 < NQ) {
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            fa[c ^ 1][i] = *reinterpret_cast<const floatx2*>(cur + fa0 + i * 32 * LDA +
                                                            4 * (q + 1));
            fb[c ^ 1][i] = *reinterpret_cast<const floatx2*>(cur + fb0 + i * 32 * LDA +
                                                            4 * (q + 1));
          }
        }
        // X slots 2q, 2q + 1 and (q < 6) dY slot q: slice step + 1 -> LDS, the
        // registers take slice step + 2
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int i = 2 * q + h;
          nxt[wrx + i * 8 * LDA] = stx[i];
          // DBG 1 (timing only): only tap group 0 reads X -- the traffic of ONE
          // shared X image
          stx[i] = buf_load(rx, (DBG == 1 && g != 0) ? kOOB : vx2 + sb, 0);
          sb += db;
        }
        if (q < NPY) {
          nxt[wry + q * 8 * LDA] = sty[q];
          sty[q] = buf_load(ry, vy2 + sa, 0);
          sa += da;
        }
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[c][i][e], fb[c][j][e],
                                                               acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  }

  // ---- output: every tap group stores its own tile ---------------------------
  if (o.out_mode == 0) {
    float* slab = a.slabs + ((size_t)split * 9 + tap) * Cout * Cin;
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
  } else {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int ci = c0 + wn * 64 + j * 32 + l31;
      if (ci >= Cin) continue;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int co = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
          if (co >= Cout) continue;
          const size_t idx = ((size_t)co * Cin + ci) * 9 + tap;
          o.dw[idx] = o.accumulate ? o.dw[idx] + acc[i][j][r] : acc[i][j][r];
        }
    }
  }
}

// ---- host side --------------------------------------------------------------
constexpr size_t kTicketSlots = 4096;  // counters per workspace (tiles per launch)
constexpr size_t kTicketPools = 256;   // distinct workspaces served per device

struct TicketPool {
  unsigned* base = nullptr;
  std::unordered_map<const void*, size_t> slice;  // workspace pointer -> slice index
  size_t next = 0;
};
std::mutex g_ticket_mu;
std::unordered_map<int, TicketPool> g_ticket_pools;  // by device

// Ticket counters live in library-owned memory that nothing else ever writes:
// zeroed once, and every launch leaves its counters zero (the last-arriving
// workgroup resets them).  Launches that may run concurrently use different
// workspaces (the slabs' existing contract), so the slice is keyed by the
// workspace pointer.  Returns nullptr when the pool does not exist yet and the
// stream is capturing (no allocation inside a capture): the caller then takes
// the slab path.
unsigned* ticket_slice(const void* workspace, hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_ticket_mu);
  TicketPool& p = g_ticket_pools[dev];
  if (!p.base) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(stream, &cap);
    if (cap != hipStreamCaptureStatusNone) return nullptr;
    const size_t bytes = kTicketPools * kTicketSlots * sizeof(unsigned);
    if (hipMalloc((void**)&p.base, bytes) != hipSuccess) {
      p.base = nullptr;
      return nullptr;
    }
    // zeroed ON the launching stream (a null-stream memset is not
    // ordered before a kernel on a non-blocking side stream); a pool that could
    // not be zeroed is released, never used as counters
    if (hipMemsetAsync(p.base, 0, bytes, stream) != hipSuccess ||
        hipStreamSynchronize(stream) != hipSuccess) {
      (void)hipFree(p.base);
      p.base = nullptr;
      return nullptr;
    }
  }
  auto it = p.slice.find(workspace);
  size_t idx;
  if (it != p.slice.end()) {
    idx = it->second;
  } else {
    // a slice is never handed to a second workspace (two live workspaces on one
    // slice could run concurrently): past the pool's capacity the caller takes
    // the slab path
    if (p.next >= kTicketPools) return nullptr;
    idx = p.next++;
    p.slice[workspace] = idx;
  }
  return p.base + idx * kTicketSlots;
}

template <int DBG>
void launch_tile_dbg(const WgradK& k, const WgradTileOut& o, int blocks, hipStream_t stream) {
  LD_LAUNCH((conv_wgrad_tile_kernel<1, 32, true, DBG>), dim3(blocks), dim3(256), 0,
                     stream, k, o);
}

template <int KG, int BK>
void launch_tile(const WgradK& k, const WgradTileOut& o, int blocks, hipStream_t stream) {
  if (KG == 1 && BK == 32) {
    // timing-attribution variants compute WRONG results: both variables needed
    const char* env = getenv("LD_ALLOW_WRONG_RESULTS") ? getenv("LD_WGRAD_DBG") : nullptr;
    if (env) {
      switch (atoi(env)) {
        case 1: return launch_tile_dbg<1>(k, o, blocks, stream);
        case 2: return launch_tile_dbg<2>(k, o, blocks, stream);
        case 4: return launch_tile_dbg<4>(k, o, blocks, stream);
        case 6: return launch_tile_dbg<6>(k, o, blocks, stream);
        case 8: return launch_tile_dbg<8>(k, o, blocks, stream);
        case 14: return launch_tile_dbg<14>(k, o, blocks, stream);
        case 15: return launch_tile_dbg<15>(k, o, blocks, stream);
        case 16: return launch_tile_dbg<16>(k, o, blocks, stream);
        case 30: return launch_tile_dbg<30>(k, o, blocks, stream);
        case 31: return launch_tile_dbg<31>(k, o, blocks, stream);
        case 32: return launch_tile_dbg<32>(k, o, blocks, stream);
        case 33: return launch_tile_dbg<33>(k, o, blocks, stream);
        case 64: return launch_tile_dbg<64>(k, o, blocks, stream);
        case 128: return launch_tile_dbg<128>(k, o, blocks, stream);
        default: break;
      }
    }
  }
  if (k.g.num_levels > 1)
    LD_LAUNCH((conv_wgrad_tile_kernel<KG, BK, true>), dim3(blocks), dim3(256 * KG),
                       0, stream, k, o);
  else
    LD_LAUNCH((conv_wgrad_tile_kernel<KG, BK, false>), dim3(blocks), dim3(256 * KG),
                       0, stream, k, o);
}

}  // namespace

bool ld_f32_wgrad_tile_cfg_ok(int kg, int bk) {
  return (bk == 32 && (kg == 1 || kg == 2 || kg == 4)) || (bk == 64 && (kg == 2 || kg == 4));
}

// 3 x 3 convs only: conv_wgrad_tap3_kernel (one workgroup = the three kw taps of a
// kernel row of a 128 x 128 tile).  256 workgroups fill the device.
int ld_f32_wgrad_tap3_launch(const WgradK& k_in, int splits, float* dw, int accumulate,
                             void* workspace, size_t workspace_bytes, hipStream_t stream,
                             int* slabs_only) {
  if (k_in.KH != 3 || k_in.KW != 3) return LD_EUNSUPPORTED;
  WgradK k = k_in;
  const int ntiles = ((k.Cout + 127) / 128) * ((k.Cin + 127) / 128) * 3;
  if (splits < 1) splits = 1;
  int jchunk = (k.J + splits - 1) / splits;
  jchunk = (jchunk + 31) / 32 * 32;
  splits = (k.J + jchunk - 1) / jchunk;
  k.splits = splits;
  k.jchunk = jchunk;
  if ((splits > 1 || slabs_only) &&
      workspace_bytes < (size_t)splits * 9 * k.Cout * k.Cin * sizeof(float))
    return LD_ENOSPACE;
  WgradTileOut o;
  o.dw = dw;
  o.accumulate = accumulate;
  o.part = nullptr;
  o.tickets = nullptr;
  o.out_mode = splits == 1 && !slabs_only ? 1 : 0;
  k.slabs = (float*)workspace;
  const int blocks = ntiles * splits;
  if (k.g.num_levels > 1 && getenv("LD_ALLOW_WRONG_RESULTS") && getenv("LD_WGRAD_DBG") &&
      atoi(getenv("LD_WGRAD_DBG")) == 1)
    LD_LAUNCH((conv_wgrad_tap3_kernel<true, 1>), dim3(blocks), dim3(768), 0, stream,
                       k, o);
  else if (k.g.num_levels > 1)
    LD_LAUNCH((conv_wgrad_tap3_kernel<true>), dim3(blocks), dim3(768), 0, stream, k,
                       o);
  else
    LD_LAUNCH((conv_wgrad_tap3_kernel<false>), dim3(blocks), dim3(768), 0, stream, k,
                       o);
  if (hipError_t e = hipGetLastError()) return (int)e;
  if (slabs_only) {
    *slabs_only = splits;
    return 0;
  }
  if (o.out_mode == 0)
    return ld_wgrad_reduce_launch(k.slabs, splits, 9, k.Cout, k.Cin, dw, accumulate, stream);
  return 0;
}

// workgroups the device holds at once for a (kg, bk) instance: LDS (2 buffers of
// 2 x 128 x (bk + 2) floats) and the 2048-thread CU limit
int ld_f32_wgrad_tile_slots(int kg, int bk) {
  const int lds = 2 * 2 * 128 * (bk + 2) * 4 + 256;
  const int by_lds = (160 * 1024) / lds;
  const int by_threads = 2048 / (256 * kg);
  // accumulators alone are 64 VGPRs: more than four waves per SIMD cannot be
  int per_cu = by_lds < by_threads ? by_lds : by_threads;
  if (per_cu * kg * 4 > 16) per_cu = 16 / (kg * 4);
  if (per_cu < 1) per_cu = 1;
  return 256 * per_cu;
}

size_t ld_f32_wgrad_tile_workspace(int Cout, int Cin, int ntaps, int splits) {
  const size_t ntiles = (size_t)((Cout + 127) / 128) * ((Cin + 127) / 128) * ntaps;
  const size_t fused = (size_t)splits * ntiles * 128 * 128 * sizeof(float);
  const size_t slabs = (size_t)splits * ntaps * Cout * Cin * sizeof(float);
  return fused > slabs ? fused : slabs;
}

// k.splits / k.jchunk are set here from cfg.  fused != 0 asks for the in-launch
// reduction; the launch falls back to slabs + the reduce launch when the ticket
// pool is unavailable.
int ld_f32_wgrad_tile_launch(const WgradK& k_in, int kg, int bk, int splits, int fused,
                             float* dw, int accumulate, void* workspace,
                             size_t workspace_bytes, hipStream_t stream, int* slabs_only) {
  if (!ld_f32_wgrad_tile_cfg_ok(kg, bk)) return LD_EUNSUPPORTED;
  WgradK k = k_in;
  const int ntaps = k.KH * k.KW;
  const int ntiles = ((k.Cout + 127) / 128) * ((k.Cin + 127) / 128) * ntaps;
  if (splits < 1) splits = 1;
  int jchunk = (k.J + splits - 1) / splits;
  jchunk = (jchunk + bk - 1) / bk * bk;
  splits = (k.J + jchunk - 1) / jchunk;  // no empty trailing split
  k.splits = splits;
  k.jchunk = jchunk;
  if (workspace_bytes < ld_f32_wgrad_tile_workspace(k.Cout, k.Cin, ntaps, splits) &&
      (splits > 1 || slabs_only))
    return LD_ENOSPACE;
  WgradTileOut o;
  o.dw = dw;
  o.accumulate = accumulate;
  o.part = (float*)workspace;
  o.tickets = nullptr;
  o.out_mode = splits == 1 && !slabs_only ? 1 : 0;
  k.slabs = (float*)workspace;
  if (splits > 1 && fused && !slabs_only && (size_t)ntiles <= kTicketSlots) {
    o.tickets = ticket_slice(workspace, stream);
    if (o.tickets) o.out_mode = 2;
  }
  const int blocks = ntiles * splits;
#define LD_WT_CASE(KG_, BK_)                                  \
  if (kg == KG_ && bk == BK_) launch_tile<KG_, BK_>(k, o, blocks, stream);
  LD_WT_CASE(1, 32)
  LD_WT_CASE(2, 32)
  LD_WT_CASE(4, 32)
  LD_WT_CASE(2, 64)
  LD_WT_CASE(4, 64)
#undef LD_WT_CASE
  if (hipError_t e = hipGetLastError()) return (int)e;
  if (slabs_only) {
    *slabs_only = splits;
    return 0;
  }
  if (o.out_mode == 0)
    return ld_wgrad_reduce_launch(k.slabs, splits, ntaps, k.Cout, k.Cin, dw, accumulate,
                                  stream);
  return 0;
}
