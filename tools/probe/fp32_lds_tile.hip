// Probe (not part of libldhip.so): fp32 1x1 convolution as a GEMM with the
// operands SHARED between the four wavefronts of a workgroup through LDS.
//
// Question it answers: the streaming kernel of
// conv.hip gives every wavefront its own 32x32 (or 64x32 ...) tile and streams
// both operands through L1 per wave -- 512 B per 32x32x2 MFMA for the 1x1 tile,
// half of the vector cache's peak per CU, which is what bounds the 50x84 / 25x42
// stages (PMC, profiles/r03_pmc_teacher_1x1_256_1024.txt).  Here a workgroup of
// 2 x 2 waves owns a (64 TM) x (64 TN) tile: every operand element is fetched
// from L1 once per WORKGROUP (half / a quarter of the bytes), staged through a
// register ring, written to one of two LDS images AFTER the step's barrier and
// read back as MFMA fragments by the two waves that need it -- the structure of
// conv_tile_c8_kernel (conv_bf16.hip), at the fp32 MFMA rate (64 cycles per
// 32x32x2: a 16-deep k-step is 512 matrix cycles per wave, twice the bf16
// kernel's budget per barrier).
//
//   Y[co][j] = sum_k Wt[k][co] * X[k][j]      Wt: [K][Cout]  X: [K][J]  Y: [Cout][J]
//
// LDS images [BK][BM] and [BK][BN] floats, k-major: a fragment read is one
// ds_read_b32 whose 32 lanes are consecutive dwords (conflict-free), the image
// rows are written 16 bytes per lane (conflict-free ds_write_b128).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr unsigned kOOB = 0x80000000u;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, bytes, 0x00020000);
}
__device__ __forceinline__ floatx4 load16(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(floatx4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}

struct GemmK {
  const float* wt;  // [K][Cout]
  const float* x;   // [K][J]
  float* y;         // [Cout][J]
  int K, Cout, J;
};

template <int TM, int TN, int BK, int NST>
__global__ __launch_bounds__(256, 2) void fp32_lds_tile_kernel(GemmK a) {
  static_assert(NST == 4, "ring slot and LDS parity are static for NST = 4");
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int A_U = BK * BM / 4 / 256;  // 16-byte units per thread
  constexpr int B_U = BK * BN / 4 / 256;
  static_assert(A_U >= 1 && B_U >= 1, "tile");
  __shared__ __attribute__((aligned(16))) float lds[2 * BK * (BM + BN)];
  float* As = lds;                // [2][BK][BM]
  float* Bs = lds + 2 * BK * BM;  // [2][BK][BN]

  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lk = lane >> 5;
  const int mtiles = (a.Cout + BM - 1) / BM;
  const int m0 = (blockIdx.x % mtiles) * BM;
  const int n0 = (blockIdx.x / mtiles) * BN;
  const int Cout = __builtin_amdgcn_readfirstlane(a.Cout);
  const int J = __builtin_amdgcn_readfirstlane(a.J);
  const int K = __builtin_amdgcn_readfirstlane(a.K);

  // this thread's 16-byte units: unit u -> row u / (BM / 4), 4 columns from 4 (u % (BM / 4))
  unsigned va[A_U], vb[B_U];
  int sa_row[A_U], sb_row[B_U];
#pragma unroll
  for (int i = 0; i < A_U; ++i) {
    const int u = t + i * 256, k = u / (BM / 4), c = 4 * (u % (BM / 4));
    sa_row[i] = k;
    va[i] = (m0 + c < Cout) ? (unsigned)(k * Cout + m0 + c) * 4u : kOOB;
  }
#pragma unroll
  for (int i = 0; i < B_U; ++i) {
    const int u = t + i * 256, k = u / (BN / 4), c = 4 * (u % (BN / 4));
    sb_row[i] = k;
    vb[i] = (n0 + c < J) ? (unsigned)(k * J + n0 + c) * 4u : kOOB;
  }
  (void)sa_row;
  (void)sb_row;
  const int nsteps = K / BK;  // host: K % BK == 0
  const rsrc_t rw = make_rsrc(a.wt, (unsigned)K * Cout * 4u);
  const rsrc_t rx = make_rsrc(a.x, (unsigned)K * J * 4u);

  floatx16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  floatx4 a_st[NST][A_U], b_st[NST][B_U];
  auto load_step = [&](int step, floatx4* ra, floatx4* rb) {
    // unconditional (no branch around ring loads); steps past the end take the
    // out-of-range voffset -- the soffset is not part of the range check
    const bool live = step < nsteps;
    const unsigned sa = (unsigned)(step * BK) * (unsigned)Cout * 4u;
    const unsigned sb = (unsigned)(step * BK) * (unsigned)J * 4u;
#pragma unroll
    for (int i = 0; i < A_U; ++i) ra[i] = load16(rw, live ? va[i] : kOOB, live ? sa : 0u);
#pragma unroll
    for (int i = 0; i < B_U; ++i) rb[i] = load16(rx, live ? vb[i] : kOOB, live ? sb : 0u);
  };
  auto store_tile = [&](int buf, const floatx4* ra, const floatx4* rb) {
#pragma unroll
    for (int i = 0; i < A_U; ++i)
      *(floatx4*)(As + buf * BK * BM + (t + i * 256) * 4) = ra[i];
#pragma unroll
    for (int i = 0; i < B_U; ++i)
      *(floatx4*)(Bs + buf * BK * BN + (t + i * 256) * 4) = rb[i];
  };
  auto mfmas = [&](int buf, int s) {
    const float* ap = As + buf * BK * BM + (2 * s + lk) * BM + wm * (BM / 2) + l31;
    const float* bp = Bs + buf * BK * BN + (2 * s + lk) * BN + wn * (BN / 2) + l31;
    float af[TM], bf[TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) af[i] = ap[i * 32];
#pragma unroll
    for (int j = 0; j < TN; ++j) bf[j] = bp[j * 32];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
  };

#pragma unroll
  for (int u = 0; u < NST - 1; ++u) load_step(u, a_st[u], b_st[u]);
  store_tile(0, a_st[0], b_st[0]);
  for (int base = 0; base < nsteps; base += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      if (base + u >= nsteps) break;
      load_step(base + u + NST - 1, a_st[(u + NST - 1) % NST], b_st[(u + NST - 1) % NST]);
      __syncthreads();  // image u complete; nobody still reads buffer (u + 1) & 1
      mfmas(u & 1, 0);
      store_tile((u + 1) & 1, a_st[(u + 1) % NST], b_st[(u + 1) % NST]);
#pragma unroll
      for (int s = 1; s < BK / 2; ++s) mfmas(u & 1, s);
    }
  }

  // epilogue: row = (r & 3) + 8 (r >> 2) + 4 lk, col = l31
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (BN / 2) + j * 32 + l31;
      if (col >= J) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * (BM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < Cout) a.y[(size_t)row * J + col] = acc[i][j][r];
      }
    }
}

#define SHAPES(X) X(1, 1, 16) X(1, 1, 32) X(2, 1, 16) X(1, 2, 16) X(2, 2, 16) X(2, 1, 32) X(1, 2, 32)

extern "C" int probe_fp32_lds_tile(const float* wt, const float* x, float* y, int K, int Cout,
                                   int J, int tm, int tn, int bk, void* stream) {
  if (K % bk != 0 || Cout % 4 != 0 || J % 4 != 0) return -1;
  GemmK a{wt, x, y, K, Cout, J};
#define CASE(TM_, TN_, BK_)                                                              \
  if (tm == TM_ && tn == TN_ && bk == BK_) {                                             \
    const int nb = ((Cout + 64 * TM_ - 1) / (64 * TM_)) * ((J + 64 * TN_ - 1) / (64 * TN_)); \
    hipLaunchKernelGGL((fp32_lds_tile_kernel<TM_, TN_, BK_, 4>), dim3(nb), dim3(256), 0,   \
                       (hipStream_t)stream, a);                                          \
    return (int)hipGetLastError();                                                       \
  }
  SHAPES(CASE)
#undef CASE
  return -2;
}
