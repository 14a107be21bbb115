// One frozen ResNet bottleneck as ONE kernel (bf16 mode, C8-only activations):
//     h1 = relu(bn1(conv1x1(x)))  ->  h2 = relu(bn2(conv3x3(h1)))  ->
//     y  = relu(bn3(conv1x1(h2)) + x)
// (mmdet/models/backbones/resnet.py:260-299, the identity blocks of the R101
// teacher's layer3: 22 of its 33 blocks, 66 of its 104 conv launches.)
//
// Why it exists: with 2 images per GPU the three convs of such a
// block are 24-28 us launches of ~1 workgroup per CU whose time is operand
// delivery, not MFMA work.  Round 4 priced the fused form on paper
// at 88 us per block and did not build it; round 5's probe
// (tools/probe/l2_weight_stream.hip, profiles/r05_probe_l2_weight_stream.json)
// measured what that price assumed: every workgroup of a launch streaming the
// block's whole 2.2 MB bf16 weight set out of L2 takes 15-23 us (46-70 B / clk /
// CU), not 88.
//
// Work decomposition.  A workgroup (4 wavefronts) owns a 4 x 16 spatial tile of
// one image = 64 output positions (50 x 84 x 2 images -> 156 workgroups, one per
// CU).  The mid activations never leave the CU:
//   G1  h1 on the tile + 1-pixel halo (6 x 18 = 108 positions, padded to 128
//       GEMM columns): [256 x 1024] x [1024 x 128].  x arrives as the C8 image,
//       128 channels at a time through a double-buffered LDS image [k8][128][8]
//       (the MFMA B layout: fragments are conflict-free ds_read_b128).  h1 ->
//       bf16 -> LDS [32][128][8]; halo positions outside the image are ZERO
//       (conv2's padding pads h1, not x).
//   G2  h2 on the 64 positions: 9 taps x [256 x 256] x [256 x 64], the B
//       fragments are shifted reads of the h1 image.  h2 -> bf16 -> LDS.
//   G3  y: [1024 x 256] x [256 x 64] in four 256-row passes; epilogue = bn3 +
//       the identity (x re-read as 8-byte C8 halves) + ReLU -> C8 image of y.
// Every wavefront owns 64 (G1, G2) / 256 (G3) weight ROWS and all columns, so a
// weight element is fetched by exactly one wavefront, straight from the bf16
// image [k8][Cout][8] into MFMA A-fragment registers (a 4-step register ring,
// no LDS, no barrier): 2.2 MB through each CU's L1 once.  The activations are
// the shared operand and live in LDS: 4 waves x 4 fragment reads per 8 MFMAs =
// half the LDS bandwidth.
//
// Numerics: the same bf16 operands enter the same fp32 accumulation order
// (tap-major, channel ascending, one accumulator per output) as the three
// conv_tile_c8_kernel launches, intermediates are rounded to bf16 exactly where
// the C8-only path rounds them -> results are bit-identical to the unfused block
// (tests/test_gpu_fused_block.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));

struct FusedK {
  const void* x;
  void* y;
  const void* w1;
  const void* w2;
  const void* w3;
  const float* s1;
  const float* b1;
  const float* s2;
  const float* b2;
  const float* s3;
  const float* b3;
  int N, H, W, P, tiles_h, tiles_w;
  unsigned x_bytes, w1_bytes, w2_bytes, w3_bytes;
};

constexpr int kTH = 4, kTW = 16;            // output tile
constexpr int kHW = kTW + 2;                // halo row length (18)
constexpr int kNH = (kTH + 2) * kHW;        // halo positions (108)
constexpr int kNQ = 128;                    // ... padded to GEMM columns
constexpr int kNP = kTH * kTW;              // output positions (64)
constexpr int kKC = 128;                    // channels of x per LDS chunk (8 k16 steps)

__device__ __forceinline__ uintx4 ldg16(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(uintx4,
                            __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

__device__ __forceinline__ floatx16 mfma(uintx4 a, uintx4 b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                 __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

template <int CIN, int MID>
__global__ __launch_bounds__(256, 1) void fused_bottleneck_c8_kernel(FusedK a) {
  static_assert(CIN == 4 * MID && MID % 64 == 0 && CIN % kKC == 0, "bottleneck widths");
  static_assert(MID == 256, "four wavefronts x 64 mid rows");
  constexpr int M8 = MID / 8;   // k8 blocks of the mid activations
  constexpr int C8 = CIN / 8;
  extern __shared__ __attribute__((aligned(16))) uintx4 lds[];
  uintx4* H1 = lds;                     // [M8][kNQ]
  uintx4* Xs = lds + M8 * kNQ;          // [2][kKC / 8][kNQ]
  uintx4* H2 = Xs;                      // [M8][kNP], after G1 (same 32 KB)
  static_assert(M8 * kNP <= 2 * (kKC / 8) * kNQ, "h2 must fit the x staging buffers");
  static_assert(kKC % 64 == 0 && (kKC / 16) % 4 == 0, "ring slots repeat per chunk");

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int l31 = lane & 31, lk = lane >> 5;
  const int tile = blockIdx.x;
  const int tw = tile % a.tiles_w;
  const int th = (tile / a.tiles_w) % a.tiles_h;
  const int n = tile / (a.tiles_w * a.tiles_h);
  const int r0 = th * kTH, c0 = tw * kTW;
  const int H = a.H, W = a.W, P = a.P;

  const rsrc_t rx = make_rsrc(a.x, a.x_bytes);
  const rsrc_t rw1 = make_rsrc(a.w1, a.w1_bytes);
  const rsrc_t rw2 = make_rsrc(a.w2, a.w2_bytes);
  const rsrc_t rw3 = make_rsrc(a.w3, a.w3_bytes);

  // halo column q -> image position (or outside)
  auto halo_pos = [&](int q, bool& ok) -> int {
    const int hr = q / kHW, hc = q - hr * kHW;
    const int ir = r0 - 1 + hr, ic = c0 - 1 + hc;
    ok = q < kNH && ir >= 0 && ir < H && ic >= 0 && ic < W;
    return ir * W + ic;
  };

  // ======================= G1: h1 = relu(bn1(W1 x)) on the halo ===============
  floatx16 acc1[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.0f;
  {
    // x staging: thread t owns halo column xq and the k8 blocks xk + 2 i of a chunk
    const int xq = t & (kNQ - 1), xk = t >> 7;
    bool xok;
    const int xp = halo_pos(xq, xok);
    const unsigned vx = xok ? (unsigned)((n * C8 + xk) * P + xp) * 16u : kOOB;
    auto load_x = [&](int c, uintx4* xr, bool live) {
#pragma unroll
      for (int i = 0; i < kKC / 16; ++i)
        xr[i] = ldg16(rx, live ? vx : kOOB, (unsigned)((c * (kKC / 8) + 2 * i) * P) * 16u);
    };
    auto store_x = [&](int buf, const uintx4* xr) {
#pragma unroll
      for (int i = 0; i < kKC / 16; ++i) Xs[(buf * (kKC / 8) + xk + 2 * i) * kNQ + xq] = xr[i];
    };
    // weights: this wave's 64 rows, lane = (row l31, k half lk) of an M tile
    unsigned va[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) va[i] = (unsigned)(lk * MID + wave * 64 + i * 32 + l31) * 16u;
    constexpr int NSTEP = CIN / 16;  // k16 steps
    uintx4 ar[4][2];
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) ar[s][i] = ldg16(rw1, va[i], (unsigned)(2 * s * MID) * 16u);
    uintx4 xr[kKC / 16];
    load_x(0, xr, true);
    store_x(0, xr);
    __syncthreads();
    constexpr int NCH = CIN / kKC;
    // Issue order is pinned with scheduling fences (hipcc otherwise sinks every
    // load below the MFMAs and waits for it at once -- ISA reading, as in
    // conv.hip's ring): per chunk  [x loads of chunk c + 1]  then per k16 step
    // [fragment reads of step s + 1] [8 MFMAs] [refill of the ring slot just used];
    // the x loads have a whole chunk (~1000 clk) to land before their LDS write, a
    // ring slot four steps.
    for (int c = 0; c < NCH; ++c) {
      load_x(c + 1, xr, c + 1 < NCH);  // no branch around the loads (out of range: zeros)
      const uintx4* xb = Xs + (c & 1) * (kKC / 8) * kNQ;
      uintx4 bf[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bf[0][j] = xb[lk * kNQ + j * 32 + l31];
      __builtin_amdgcn_sched_barrier(0);
      constexpr int SPC = kKC / 16;  // k16 steps per chunk
#pragma unroll
      for (int s = 0; s < SPC; ++s) {
        if (s + 1 < SPC) {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            bf[(s + 1) & 1][j] = xb[(2 * (s + 1) + lk) * kNQ + j * 32 + l31];
        }
        const int kn = c * SPC + s + 4;  // the step this ring slot serves next
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          acc1[0][j] = mfma(ar[s & 3][0], bf[s & 1][j], acc1[0][j]);
          acc1[1][j] = mfma(ar[s & 3][1], bf[s & 1][j], acc1[1][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // refilled IN PLACE behind its MFMAs: the slot is the oldest load in flight
        // when it is needed again (exact vmcnt, no wait for younger slots)
#pragma unroll
        for (int i = 0; i < 2; ++i)
          ar[s & 3][i] =
              ldg16(rw1, kn < NSTEP ? va[i] : kOOB, (unsigned)(2 * kn * MID) * 16u);
        __builtin_amdgcn_sched_barrier(0);
      }
      store_x((c + 1) & 1, xr);
      __syncthreads();
    }
    // epilogue: affine + ReLU, zero outside the image, bf16, into the h1 image
    bool qok[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) (void)halo_pos(j * 32 + l31, qok[j]);
    // every load of the epilogue is issued before the first use (one exposed
    // round trip instead of eight: a single wavefront per SIMD hides nothing)
    floatx4_t sc1[2][4], sh1[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row0 = wave * 64 + i * 32 + 8 * g + 4 * lk;
        sc1[i][g] = *reinterpret_cast<const floatx4_t*>(a.s1 + row0);
        sh1[i][g] = *reinterpret_cast<const floatx4_t*>(a.b1 + row0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row0 = wave * 64 + i * 32 + 8 * g + 4 * lk;
        const floatx4_t sc = sc1[i][g], sh = sh1[i][g];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          floatx4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            float u = acc1[i][j][4 * g + e] * sc[e] + (sh[e] + 0.0f);
            u = fmaxf(u, 0.0f);
            v[e] = qok[j] ? u : 0.0f;
          }
          reinterpret_cast<uintx2*>(H1)[((row0 >> 3) * kNQ + j * 32 + l31) * 2 + lk] =
              __builtin_bit_cast(uintx2, __builtin_convertvector(v, bf16x4));
        }
      }
  }
  __syncthreads();

  // ======================= G2: h2 = relu(bn2(W2 * h1)), 3 x 3 =================
  floatx16 acc2[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[i][j][r] = 0.0f;
  {
    int qb[2];  // halo index of this lane's output position, tap (0, 0)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int np = j * 32 + l31;
      qb[j] = (np >> 4) * kHW + (np & 15);
    }
    unsigned va[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) va[i] = (unsigned)(lk * MID + wave * 64 + i * 32 + l31) * 16u;
    constexpr int NSTEP = 9 * MID / 16;  // 144: the image is contiguous in k8 across taps
    // G2 / G3 steps are 4 MFMAs (128 clk) long: an 8-slot ring keeps a weight
    // fragment ~1000 clk in flight, the L2 latency under load (a 4-slot ring, 512
    // clk, stalled every step: the first version ran at 54 us per block)
    constexpr int RD = 8;
    static_assert(NSTEP % RD == 0 && (MID / 16) % RD == 0, "ring groups per tap");
    uintx4 ar[RD][2];
#pragma unroll
    for (int s = 0; s < RD; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) ar[s][i] = ldg16(rw2, va[i], (unsigned)(2 * s * MID) * 16u);
    auto h1_frag = [&](int st, int j) -> uintx4 {  // B fragment of flat step st
      const int tap = st / (MID / 16), kk = st - tap * (MID / 16);
      const int kh = tap / 3, kw = tap - kh * 3;
      return H1[(2 * kk + lk) * kNQ + qb[j] + kh * kHW + kw];
    };
    uintx4 bf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[0][j] = h1_frag(0, j);
    for (int g8 = 0; g8 < NSTEP / RD; ++g8) {
#pragma unroll
      for (int s = 0; s < RD; ++s) {
        const int st = g8 * RD + s;
        const int sn = st + 1 < NSTEP ? st + 1 : st;  // the last prefetch is a repeat
#pragma unroll
        for (int j = 0; j < 2; ++j) bf[(s + 1) & 1][j] = h1_frag(sn, j);
        const int kn = st + RD;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc2[0][j] = mfma(ar[s][0], bf[s & 1][j], acc2[0][j]);
          acc2[1][j] = mfma(ar[s][1], bf[s & 1][j], acc2[1][j]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
          ar[s][i] = ldg16(rw2, kn < NSTEP ? va[i] : kOOB, (unsigned)(2 * kn * MID) * 16u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    floatx4_t sc2[2][4], sh2[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row0 = wave * 64 + i * 32 + 8 * g + 4 * lk;
        sc2[i][g] = *reinterpret_cast<const floatx4_t*>(a.s2 + row0);
        sh2[i][g] = *reinterpret_cast<const floatx4_t*>(a.b2 + row0);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row0 = wave * 64 + i * 32 + 8 * g + 4 * lk;
        const floatx4_t sc = sc2[i][g], sh = sh2[i][g];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          floatx4_t v;
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = fmaxf(acc2[i][j][4 * g + e] * sc[e] + (sh[e] + 0.0f), 0.0f);
          reinterpret_cast<uintx2*>(H2)[((row0 >> 3) * kNP + j * 32 + l31) * 2 + lk] =
              __builtin_bit_cast(uintx2, __builtin_convertvector(v, bf16x4));
        }
      }
  }
  __syncthreads();

  // ======================= G3: y = relu(bn3(W3 h2) + x) =======================
  {
    // this lane's two output positions
    // byte offset of (n, c8 = 0, p) in a C8 image + the lane's half; a position
    // outside the image gets the out-of-range offset (loads 0, stores dropped: no
    // branch, no exec mask in the pipelined region)
    unsigned pbase[2];
    const rsrc_t ry = make_rsrc(a.y, a.x_bytes);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int np = j * 32 + l31;
      const int ir = r0 + (np >> 4), ic = c0 + (np & 15);
      pbase[j] = ir < H && ic < W
                     ? (unsigned)(((size_t)n * C8 * P + (size_t)(ir * W + ic)) * 16 + lk * 8)
                     : kOOB;
    }
    constexpr int NSTEP = 4 * (MID / 16);  // 4 passes of 64 rows x 16 k16 steps
    auto va3 = [&](int f, int i) -> unsigned {  // weight row of flat step f
      const int mc = f / (MID / 16);
      return (unsigned)(lk * CIN + wave * 256 + mc * 64 + i * 32 + l31) * 16u;
    };
    auto so3 = [&](int f) -> unsigned {
      return (unsigned)(2 * (f % (MID / 16)) * CIN) * 16u;
    };
    constexpr int RD = 8;
    static_assert((MID / 16) % RD == 0, "ring groups per pass");
    uintx4 ar[RD][2];
#pragma unroll
    for (int s = 0; s < RD; ++s)
#pragma unroll
      for (int i = 0; i < 2; ++i) ar[s][i] = ldg16(rw3, va3(s, i), so3(s));
    floatx16 acc3[2][2];
    uintx4 bf[2][2];
#pragma unroll
    for (int j = 0; j < 2; ++j) bf[0][j] = H2[lk * kNP + j * 32 + l31];
    constexpr int GPP = MID / 16 / RD;  // ring groups per pass (2)
    for (int mc = 0; mc < 4; ++mc) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc3[i][j][r] = 0.0f;
      // the pass's epilogue operands (bn3 coefficients, the identity as 8-byte C8
      // halves) are requested under its LAST ring group: older than that group's
      // refills in the load queue, they have landed when the epilogue starts
      floatx4_t sc3[2][4], sh3[2][4];
      uintx2 rr[2][4][2];
#pragma unroll
      for (int gp = 0; gp < GPP; ++gp) {
        const int kk0 = gp * RD;
        if (gp == GPP - 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const int row0 = wave * 256 + mc * 64 + i * 32 + 8 * g + 4 * lk;
              const unsigned crow = (unsigned)(row0 >> 3) * (unsigned)P * 16u;
              sc3[i][g] = *reinterpret_cast<const floatx4_t*>(a.s3 + row0);
              sh3[i][g] = *reinterpret_cast<const floatx4_t*>(a.b3 + row0);
#pragma unroll
              for (int j = 0; j < 2; ++j)
                rr[i][g][j] = __builtin_bit_cast(
                    uintx2, __builtin_amdgcn_raw_buffer_load_b64(rx, pbase[j], crow, 0));
            }
          __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < RD; ++s) {
          const int kn1 = (kk0 + s + 1) % (MID / 16);  // next step's k (wraps into the next pass)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            bf[(s + 1) & 1][j] = H2[(2 * kn1 + lk) * kNP + j * 32 + l31];
          const int fn = (mc * GPP + gp) * RD + s + RD;
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc3[0][j] = mfma(ar[s][0], bf[s & 1][j], acc3[0][j]);
            acc3[1][j] = mfma(ar[s][1], bf[s & 1][j], acc3[1][j]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 2; ++i)
            ar[s][i] = ldg16(rw3, fn < NSTEP ? va3(fn, i) : kOOB, fn < NSTEP ? so3(fn) : 0u);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      // pass complete: bn3 + identity + ReLU -> the C8 image of y
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row0 = wave * 256 + mc * 64 + i * 32 + 8 * g + 4 * lk;
          const unsigned crow = (unsigned)(row0 >> 3) * (unsigned)P * 16u;
          const floatx4_t sc = sc3[i][g], sh = sh3[i][g];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const floatx4_t rq =
                __builtin_convertvector(__builtin_bit_cast(bf16x4, rr[i][g][j]), floatx4_t);
            floatx4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              float u = acc3[i][j][4 * g + e] * sc[e] + (sh[e] + 0.0f);
              u += rq[e];
              v[e] = fmaxf(u, 0.0f);
            }
            __builtin_amdgcn_raw_buffer_store_b64(
                __builtin_bit_cast(uintx2, __builtin_convertvector(v, bf16x4)), ry, pbase[j],
                crow, 0);
          }
        }
    }
  }
}

constexpr size_t kFusedLds = ((size_t)(256 / 8) * kNQ + 2 * (kKC / 8) * kNQ) * 16;  // 128 KB

}  // namespace

extern "C" int ld_bottleneck_c8_supported(int Cin, int mid, int H, int W) {
  return Cin == 1024 && mid == 256 && H >= 2 && W >= 2 ? 1 : 0;
}

extern "C" int ld_bottleneck_c8_forward(const ld_bottleneck_t* b, const void* x_c8,
                                        void* y_c8, ld_stream_t stream) {
  if (!b || !x_c8 || !y_c8 || x_c8 == y_c8) return LD_EINVAL;
  if (!b->w1 || !b->w2 || !b->w3 || !b->scale1 || !b->shift1 || !b->scale2 ||
      !b->shift2 || !b->scale3 || !b->shift3 || b->N < 1)
    return LD_EINVAL;
  if (!ld_bottleneck_c8_supported(b->Cin, b->mid, b->H, b->W)) return LD_EUNSUPPORTED;
  const size_t xb = (size_t)b->N * b->Cin * b->H * b->W * 2;
  if (xb >= (size_t)kOOB) return LD_EUNSUPPORTED;
  FusedK k{};
  k.x = x_c8;
  k.y = y_c8;
  k.w1 = b->w1; k.w2 = b->w2; k.w3 = b->w3;
  k.s1 = b->scale1; k.b1 = b->shift1;
  k.s2 = b->scale2; k.b2 = b->shift2;
  k.s3 = b->scale3; k.b3 = b->shift3;
  k.N = b->N; k.H = b->H; k.W = b->W; k.P = b->H * b->W;
  k.tiles_h = (b->H + kTH - 1) / kTH;
  k.tiles_w = (b->W + kTW - 1) / kTW;
  k.x_bytes = (unsigned)xb;
  k.w1_bytes = (unsigned)((size_t)b->Cin * b->mid * 2);
  k.w2_bytes = (unsigned)((size_t)9 * b->mid * b->mid * 2);
  k.w3_bytes = (unsigned)((size_t)b->mid * b->Cin * 2);
  auto kern = fused_bottleneck_c8_kernel<1024, 256>;
  static const hipError_t attr = hipFuncSetAttribute(
      (const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFusedLds);
  if (attr != hipSuccess) return (int)attr;
  const int blocks = b->N * k.tiles_h * k.tiles_w;
  LD_LAUNCH(kern, dim3(blocks), dim3(256), kFusedLds, (hipStream_t)stream, k);
  return (int)hipGetLastError();
}
