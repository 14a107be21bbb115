// Shared between conv.hip (fp32 MFMA kernels) and conv_bf16.hip (bf16 MFMA
// kernels): the kernel-side conv descriptor, gather geometry, buffer-load
// helpers and the shape-tuning table.  Everything here has internal linkage
// except the tune table accessors, which conv.hip defines once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ld_hip.h"
#include "ld_launch.h"

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
  int Kpad;               // rows per tap of the weight image (Cin rounded up)
  int pipe;               // use the software-pipelined main loop
  int x_c8;               // x is the bf16 channel-blocked image (N, Cin/8, Pin, 8)
  void* y_c8;             // optional: also write y as (N, Cout/8, Pout, 8) bf16
  const void* res_c8;     // optional: residual as a C8 image (then residual == null)
  float* y_raw;           // optional second output: acc (+ bias) before the affine
  void* raw_c8;           // ... or that output as a bf16 C8 image only (C8 kernels)
  unsigned x_bytes, wt_bytes;  // buffer-descriptor extents
  // MODE 1 (data-gradient of a stride-2 conv, one output-parity class per
  // launch): g.lv[].Hout/Wout/off_out describe the COMPACT grid of the class;
  // the class is (ph, pw); only taps kh = kh0 + 2*i (i < nth), kw = kw0 + 2*j
  // (j < ntw) reach it; input row = hc + ch0 + i, col = wc + cw0 + j.
  int ph, pw, kh0, kw0, nth, ntw, ch0, cw0;
  int Pfull;                       // positions per (n, c) row of the output
  int tune_plain;                  // shape-table key ignores the residual (dgrad + addend)
  int fW[LD_MAX_LEVELS], foff[LD_MAX_LEVELS];  // full output row length/offset
  Geo g;
};


struct WgradK {
  const float* x;    // (N, Cin, Pin)
  const float* dy;   // (N, Cout, Pout)
  float* slabs;      // [split][tap][Cout][Cin]
  int N, Cin, Cout, KH, KW;
  int Pin, Pout;
  int J, splits, jchunk;  // jchunk: columns per split (multiple of the j-step)
  unsigned x_bytes, dy_bytes;
  Geo g;
};

// ---- cross-file hooks (external linkage) -------------------------------------
// conv_bf16.hip: bf16-MFMA streaming forward / data-gradient launch + tuning
// (mode 0 = conv, 1 = stride-2 dgrad parity class), wave-private bf16 wgrad.
int ld_bf16_stream_launch(int mode, const ConvK& k, hipStream_t stream);
int ld_bf16_wgrad_c8_launch(const WgradK& k, hipStream_t stream);
bool ld_bf16_wgrad_c8_tiled(int Cout, int Cin);  // 128 x 128 workgroup tiles?
int ld_bf16_wgrad_c8_tile_splits(int Cout, int Cin, int ntaps, int J);
int ld_bf16_stream_tune(int mode, const ConvK& k, hipStream_t stream);
// conv_t256.hip: C8 operands, 8-wave workgroup tiles (BM x BN) fed by LDS-DMA
bool ld_bf16_t256_fits(const ConvK& k, int BM, int BN);
int ld_bf16_t256_launch(int mode, const ConvK& k, int BM, int BN, hipStream_t stream);
int ld_bf16_wgrad_launch(const WgradK& k, hipStream_t stream);
bool ld_bf16_wgrad_tiled(int Cout, int Cin, int Pout);     // which bf16 wgrad kernel
int ld_bf16_wgrad_splits(int Cout, int Cin, int ntaps, int J);  // its j-split count
// conv_wgrad.hip: fp32 workgroup-tiled weight gradient (128 x 128 tiles, kg
// k-groups of four waves, bk columns per staged slice, `splits` workgroups per
// tile; fused = combine the splits inside the launch).  Sets k.splits / k.jchunk
// itself; issues the slab reduce launch when the combination is not fused.
bool ld_f32_wgrad_tile_cfg_ok(int kg, int bk);
int ld_f32_wgrad_tile_slots(int kg, int bk);  // workgroups resident on the device
size_t ld_f32_wgrad_tile_workspace(int Cout, int Cin, int ntaps, int splits);
// slabs_only (round 5, deferred reduction): write the split partials as slabs
// [split][tap][Cout][Cin] into the workspace even for one split, issue NO reduce
// launch, and report the number of slabs written through *slabs_only.
int ld_f32_wgrad_tile_launch(const WgradK& k, int kg, int bk, int splits, int fused,
                             float* dw, int accumulate, void* workspace,
                             size_t workspace_bytes, hipStream_t stream,
                             int* slabs_only = nullptr);
int ld_f32_wgrad_tap3_launch(const WgradK& k, int splits, float* dw, int accumulate,
                             void* workspace, size_t workspace_bytes, hipStream_t stream,
                             int* slabs_only = nullptr);
// conv.hip: fixed-order sum of the wgrad slabs into dW (shared by both families)
int ld_wgrad_reduce_launch(const float* slabs, int splits, int ntaps, int Cout,
                           int Cin, float* dw, int accumulate, hipStream_t stream);

namespace {
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;
constexpr int WBK = 32;   // wgrad k-slice (spatial positions) per step
constexpr int WLD = WBK + 1;  // odd LDS row stride -> conflict-free columns

// Weight-image rows per tap are padded with zero rows to a multiple of this,
// so the k-tail of the GEMM needs no masking on the A side.
constexpr int kKPad = 32;
inline int kpad_rows(int k) { return (k + kKPad - 1) / kKPad * kKPad; }

// A voffset at/above this is out of range for every descriptor we build
// (extents are checked < 2 GiB on the host): buffer loads return 0 there.
constexpr unsigned kOOB = 0x80000000u;

typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) {
  // descriptor inputs pinned wave-uniform (cdna_hip_programming.md T20)
  const uintptr_t u = (uintptr_t)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
  const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
  void* q = (void*)(((uintptr_t)hi << 32) | lo);
  return __builtin_amdgcn_make_buffer_rsrc(q, 0, __builtin_amdgcn_readfirstlane(bytes),
                                           0x00020000);
}

__device__ __forceinline__ float buf_load(rsrc_t r, unsigned voff, unsigned soff) {
  return __builtin_bit_cast(float,
                            __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}

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

}  // namespace

// ---- shape-tuning table (one instance, defined in conv.hip) ----------------
// key: 18 ints = {MODE, Cin, Cout, KH, KW, stride, pad, J, num_levels, Hin0,
// Win0, ph, pw, relu, has_residual, has_affine, family, 0}; family 0 = fp32
// streaming kernels, 1 = bf16 streaming kernels.  value: {tm, tn, wvm, d, ks
// [, cap]}.
struct LdTuneKey {
  int v[18];
};
struct LdTuneCfg {
  int tm, tn, wvm, d, ks;
  // fp32 streaming family only: workgroups per CU the launch is held to through
  // dynamic LDS (0 = whatever the registers allow).  See launch_stream_cfg.
  int cap = 0;
};
bool ld_tune_lookup(const LdTuneKey& key, LdTuneCfg* out);
void ld_tune_store(const LdTuneKey& key, const LdTuneCfg& cfg);

namespace {
inline LdTuneKey make_tune_key(int mode, int family, const ConvK& k) {
  // a residual is a residual whether it arrives as fp32 or as a C8 image (round 6: the
  // trainable trunk hands C8 images on, and its conv3 launches must keep finding their
  // tuned rows); LD_TUNE_KEY_RESC8=0: the key of rounds 2-5 (fp32 residual only)
  static const bool resc8 = [] {
    const char* e = getenv("LD_TUNE_KEY_RESC8");
    return !(e && e[0] == '0');
  }();
  const bool res = k.residual != nullptr || (resc8 && k.res_c8 != nullptr);
  return LdTuneKey{{mode, k.Cin, k.Cout, k.KH, k.KW, k.g.stride, k.g.pad, k.J,
                    k.g.num_levels, k.g.lv[0].Hin, k.g.lv[0].Win, k.ph, k.pw, k.relu,
                    res && !k.tune_plain, k.scale != nullptr, family, 0}};
}
}  // namespace
