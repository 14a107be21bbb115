// Weight images for ALL trainable convs in one launch, tiled (round 5).
// After every optimizer step the (Cout, Cin, KH, KW) parameters are rewritten into
// the GEMM images -- fp32 [tap][Cin_pad][Cout] / [flipped tap][Cout_pad][Cin]
// (conv.hip) or bf16 [tap][K/8][C][8] (conv_bf16.hip).  The round-2 batch kernels
// compute one image element per thread: coalesced writes, but each read of the
// forward image strides over Cin * ntaps floats -- 250 us per step for 128 MB of
// parameters (1.5 TB/s), on the critical path between the optimizer and the next
// forward.  Here a workgroup owns a 32 (co) x 32 (ci) x ntaps tile: coalesced reads of
// 32 * ntaps floats per output channel into LDS, then 128-byte (fp32) / 512-byte
// (bf16) row segments of both images.  Same values bit for bit (pure data movement;
// bf16: the same round-to-nearest-even conversion).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "conv_common.h"

namespace {

constexpr int kT = 32;        // tile edge (channels)
constexpr int kMaxTaps = 9;   // 1x1 and 3x3 convs (others: the per-element kernels)

// reduction extents of the images: fp32 rows padded to kKPad (conv_common.h), bf16 to 16
__host__ __device__ inline int wt_pad(int k, bool bf16) {
  return bf16 ? (k + 15) / 16 * 16 : (k + 31) / 32 * 32;
}
static_assert(kKPad == 32, "wt_pad mirrors kpad_rows");

typedef __bf16 wt_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wt_floatx8 __attribute__((ext_vector_type(8)));
typedef unsigned wt_uintx4 __attribute__((ext_vector_type(4)));

template <bool BF16>
__global__ __launch_bounds__(256) void weight_transform_tiled_kernel(
    const ld_wt_job_t* __restrict__ jobs, const int32_t* __restrict__ block_job) {
  __shared__ float tile[kT][kT * kMaxTaps + 1];
  const ld_wt_job_t j = jobs[block_job[blockIdx.x]];
  const int ntaps = j.ntaps, Cout = j.Cout, Cin = j.Cin;
  // padded extents of the reduction dimension of each image
  const int cinp = wt_pad(Cin, BF16), coutp = wt_pad(Cout, BF16);
  const int tci = (max(cinp, Cin) + kT - 1) / kT;
  const int b = blockIdx.x - j.first_block;
  const int co0 = (b / tci) * kT, ci0 = (b % tci) * kT;
  const int t = threadIdx.x;
  // ---- load: row r = output channel co0 + r, 32 * ntaps consecutive floats
  const int rowlen = kT * ntaps;
  for (int idx = t; idx < kT * rowlen; idx += 256) {
    const int r = idx / rowlen, q = idx - r * rowlen;  // q = ci_local * ntaps + tap
    const int co = co0 + r, ci = ci0 + q / ntaps;
    float v = 0.0f;
    if (co < Cout && ci < Cin) v = j.w[((size_t)co * Cin + ci0) * ntaps + q];
    tile[r][q] = v;
  }
  __syncthreads();
  if (!BF16) {
    // fwd [tap][ci (cinp rows)][co]: rows of 32 consecutive co
    if (j.wt_fwd) {
      const int co = co0 + (t & 31);
      for (int rr = t >> 5; rr < kT * ntaps; rr += 8) {
        const int tap = rr / kT, cl = rr - tap * kT, ci = ci0 + cl;
        if (ci < cinp && co < Cout)
          j.wt_fwd[((size_t)tap * cinp + ci) * Cout + co] = tile[t & 31][cl * ntaps + tap];
      }
    }
    // bwd [ntaps - 1 - tap][co (coutp rows)][ci]: rows of 32 consecutive ci
    if (j.wt_bwd) {
      const int ci = ci0 + (t & 31);
      for (int rr = t >> 5; rr < kT * ntaps; rr += 8) {
        const int tap = rr / kT, rl = rr - tap * kT, co = co0 + rl;
        if (co < coutp && ci < Cin)
          j.wt_bwd[((size_t)(ntaps - 1 - tap) * coutp + co) * Cin + ci] =
              tile[rl][(t & 31) * ntaps + tap];
      }
    }
  } else {
    // fwd bf16 [tap][cinp / 8][Cout][8]: one 16-byte unit per (tap, k block, co)
    if (j.wt_fwd) {
      wt_uintx4* out = reinterpret_cast<wt_uintx4*>(j.wt_fwd);
      for (int u = t; u < ntaps * (kT / 8) * kT; u += 256) {
        const int cl = u % kT, kb = (u / kT) % (kT / 8), tap = u / (kT * (kT / 8));
        const int co = co0 + cl, ci8 = ci0 / 8 + kb;
        if (co >= Cout || ci8 * 8 >= cinp) continue;
        wt_floatx8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = tile[cl][(kb * 8 + e) * ntaps + tap];
        out[((size_t)tap * (cinp / 8) + ci8) * Cout + co] =
            __builtin_bit_cast(wt_uintx4, __builtin_convertvector(f, wt_bf16x8));
      }
    }
    // bwd bf16 [ntaps - 1 - tap][coutp / 8][Cin][8]
    if (j.wt_bwd) {
      wt_uintx4* out = reinterpret_cast<wt_uintx4*>(j.wt_bwd);
      for (int u = t; u < ntaps * (kT / 8) * kT; u += 256) {
        const int cl = u % kT, kb = (u / kT) % (kT / 8), tap = u / (kT * (kT / 8));
        const int ci = ci0 + cl, co8 = co0 / 8 + kb;
        if (ci >= Cin || co8 * 8 >= coutp) continue;
        wt_floatx8 f;
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = tile[kb * 8 + e][cl * ntaps + tap];
        out[((size_t)(ntaps - 1 - tap) * (coutp / 8) + co8) * Cin + ci] =
            __builtin_bit_cast(wt_uintx4, __builtin_convertvector(f, wt_bf16x8));
      }
    }
  }
}

}  // namespace

// Blocks a job owns in the tiled launches below: 32 x 32 channel tiles over the
// PADDED extents (the pad rows of an image are zero-filled by the tiles covering
// them); 0 = this weight is not served (more than 9 taps): use the per-element
// batch entry point.
extern "C" int ld_conv_weight_transform_tiles(int Cout, int Cin, int ntaps, int bf16) {
  if (Cout < 1 || Cin < 1 || ntaps < 1 || ntaps > kMaxTaps) return 0;
  const int cinp = wt_pad(Cin, bf16 != 0), coutp = wt_pad(Cout, bf16 != 0);
  return ((coutp + kT - 1) / kT) * ((cinp + kT - 1) / kT);
}

extern "C" int ld_conv_weight_transform_batch_tiled(const ld_wt_job_t* jobs,
                                                    const int32_t* block_job, int nblocks,
                                                    int bf16, ld_stream_t stream) {
  if (!jobs || !block_job || nblocks < 1) return LD_EINVAL;
  if (bf16)
    LD_LAUNCH(weight_transform_tiled_kernel<true>, dim3(nblocks), dim3(256), 0,
              (hipStream_t)stream, jobs, block_job);
  else
    LD_LAUNCH(weight_transform_tiled_kernel<false>, dim3(nblocks), dim3(256), 0,
              (hipStream_t)stream, jobs, block_job);
  return (int)hipGetLastError();
}
