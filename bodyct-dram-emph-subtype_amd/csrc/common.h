// common.h -- shared helpers for the gfx950 kernels of libdram_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include "dram_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DRAM_LAUNCH_CHECK()                         \
  do {                                              \
    hipError_t _e = hipGetLastError();              \
    if (_e != hipSuccess) return (int)_e;           \
  } while (0)

// Kernel timeline (profile.hip): `DramProf p(family, variant, executed MFMA flops, algorithmic HBM bytes,
// stream);` in front of a launch brackets it with two events when recording is on (one bool test when off).
extern bool g_dram_prof_on;
void dram_prof_begin(int family, int variant, double mfma_flops, double hbm_bytes, double alg_flops, hipStream_t s);
void dram_prof_end(hipStream_t s);
struct DramProf {
  const bool on;
  hipStream_t s;
  // alg_flops: direct-convolution FLOPs the launch stands for (default: what it executes)
  DramProf(int family, int variant, double mfma_flops, double hbm_bytes, hipStream_t st, double alg_flops = -1.0)
      : on(g_dram_prof_on), s(st) {
    if (on) dram_prof_begin(family, variant, mfma_flops, hbm_bytes, alg_flops < 0.0 ? mfma_flops : alg_flops, st);
  }
  ~DramProf() {
    if (on) dram_prof_end(s);
  }
  DramProf(const DramProf&) = delete;
};

// ---------------------------------------------------------------------------------------------------------
// Storage types of activation-sized tensors: float (the default path: the reference's arithmetic) or bf16
// (`--precision bf16` of the reference's Lightning trainer, train.py:46; BASELINE configs[2], [4]).  The
// element-wise kernels are templates over the storage type T and do all arithmetic in fp32: ld4 / st4 move
// four channels of one voxel (16 B as float4, 8 B as four bf16), offsets in ELEMENTS, multiples of 4.
typedef unsigned short bf16_t;   // raw bits; conversions below (round to nearest even, NaN stays NaN)

__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float((unsigned)h << 16); }
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  bf16x2_t v;                                  // plain casts: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN-preserving)
  v[0] = (__bf16)lo;
  v[1] = (__bf16)hi;
  return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack2_bf16(f, 0.f) & 0xffffu); }

template <typename T> __device__ __forceinline__ float4 ld4(const T* p, long off);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p, long off) {
  return *reinterpret_cast<const float4*>(p + off);
}
template <> __device__ __forceinline__ float4 ld4<bf16_t>(const bf16_t* p, long off) {
  const uint2 u = *reinterpret_cast<const uint2*>(p + off);
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4(T* p, long off, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, long off, float4 v) {
  *reinterpret_cast<float4*>(p + off) = v;
}
template <> __device__ __forceinline__ void st4<bf16_t>(bf16_t* p, long off, float4 v) {
  *reinterpret_cast<uint2*>(p + off) = make_uint2(pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w));
}
template <typename T> __device__ __forceinline__ float ld1(const T* p, long off);
template <> __device__ __forceinline__ float ld1<float>(const float* p, long off) { return p[off]; }
template <> __device__ __forceinline__ float ld1<bf16_t>(const bf16_t* p, long off) { return bf16_to_f32(p[off]); }
template <typename T> __device__ __forceinline__ void st1(T* p, long off, float v);
template <> __device__ __forceinline__ void st1<float>(float* p, long off, float v) { p[off] = v; }
template <> __device__ __forceinline__ void st1<bf16_t>(bf16_t* p, long off, float v) { p[off] = f32_to_bf16(v); }

// VW channels of one voxel per thread, 16 B when VW * sizeof(T) == 16 (fp32: 4, bf16: 8): the gather kernels of
// pool_up.hip cost one address per lane and instruction whatever its width, so bf16 storage moves 8 channels per lane.
template <int VW> struct fvec { float v[VW]; };
template <typename T, int VW> __device__ __forceinline__ fvec<VW> ldv(const T* p, long off);
template <> __device__ __forceinline__ fvec<4> ldv<float, 4>(const float* p, long off) {
  const float4 t = *reinterpret_cast<const float4*>(p + off);
  return fvec<4>{{t.x, t.y, t.z, t.w}};
}
template <> __device__ __forceinline__ fvec<4> ldv<bf16_t, 4>(const bf16_t* p, long off) {
  const uint2 u = *reinterpret_cast<const uint2*>(p + off);
  return fvec<4>{{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                  __uint_as_float(u.y & 0xffff0000u)}};
}
template <> __device__ __forceinline__ fvec<8> ldv<bf16_t, 8>(const bf16_t* p, long off) {
  const uint4 u = *reinterpret_cast<const uint4*>(p + off);
  return fvec<8>{{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                  __uint_as_float(u.y & 0xffff0000u), __uint_as_float(u.z << 16), __uint_as_float(u.z & 0xffff0000u),
                  __uint_as_float(u.w << 16), __uint_as_float(u.w & 0xffff0000u)}};
}
template <typename T, int VW> __device__ __forceinline__ void stv(T* p, long off, const fvec<VW>& a);
template <> __device__ __forceinline__ void stv<float, 4>(float* p, long off, const fvec<4>& a) {
  *reinterpret_cast<float4*>(p + off) = make_float4(a.v[0], a.v[1], a.v[2], a.v[3]);
}
template <> __device__ __forceinline__ void stv<bf16_t, 4>(bf16_t* p, long off, const fvec<4>& a) {
  *reinterpret_cast<uint2*>(p + off) = make_uint2(pack2_bf16(a.v[0], a.v[1]), pack2_bf16(a.v[2], a.v[3]));
}
template <> __device__ __forceinline__ void stv<bf16_t, 8>(bf16_t* p, long off, const fvec<8>& a) {
  *reinterpret_cast<uint4*>(p + off) = make_uint4(pack2_bf16(a.v[0], a.v[1]), pack2_bf16(a.v[2], a.v[3]),
                                                  pack2_bf16(a.v[4], a.v[5]), pack2_bf16(a.v[6], a.v[7]));
}

// Streaming forms (non-temporal cache policy) for tensors a kernel touches exactly once: with one-shot blocks (below)
// an element-wise pass moves 6.2 TB/s read+write instead of 4.6-5.0 (tools/stream_probe.hip).
typedef float f32x4_nt __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_nt __attribute__((ext_vector_type(2)));
template <typename T> __device__ __forceinline__ float4 ld4s(const T* p, long off);
template <> __device__ __forceinline__ float4 ld4s<float>(const float* p, long off) {
  const f32x4_nt t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_nt*>(p + off));
  return make_float4(t.x, t.y, t.z, t.w);
}
template <> __device__ __forceinline__ float4 ld4s<bf16_t>(const bf16_t* p, long off) {
  const u32x2_nt u = __builtin_nontemporal_load(reinterpret_cast<const u32x2_nt*>(p + off));
  return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16),
                     __uint_as_float(u.y & 0xffff0000u));
}
template <typename T> __device__ __forceinline__ void st4s(T* p, long off, float4 v);
template <> __device__ __forceinline__ void st4s<float>(float* p, long off, float4 v) {
  const f32x4_nt t = {v.x, v.y, v.z, v.w};
  __builtin_nontemporal_store(t, reinterpret_cast<f32x4_nt*>(p + off));
}
template <> __device__ __forceinline__ void st4s<bf16_t>(bf16_t* p, long off, float4 v) {
  const u32x2_nt t = {pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w)};
  __builtin_nontemporal_store(t, reinterpret_cast<u32x2_nt*>(p + off));
}

// Test / A-B switches (DRAM_CONV_ALGO, DRAM_W2D_V, DRAM_WINO_TILING, DRAM_EW_SHAPE, ...) are read from the environment
// ONLY when DRAM_TUNING=1 is set as well (tests/conftest.py and the tools set it): a stray DRAM_* variable in a user's
// environment cannot silently change which kernel variant the product path runs.
static inline const char* tune_env(const char* name) {
  static const bool on = [] { const char* t = getenv("DRAM_TUNING"); return t && t[0] == '1'; }();
  return on ? getenv(name) : nullptr;
}

// Launch shape of the streaming element-wise kernels: ONE-SHOT blocks (block b owns elements [b * per, (b + 1) * per),
// no grid-stride loop) -- the dispatcher hands blocks out in order, so the chip sweeps a narrow moving window of each
// tensor; a capped grid with a grid-stride loop reads + writes 1 GiB at 4.6-5.0 TB/s, one-shot blocks at 6.0-6.2
// (tools/stream_probe.hip).  DRAM_EW_SHAPE=0 restores the capped grid (A/B).
static inline int ew_shape() {
  static const int v = tune_env("DRAM_EW_SHAPE") ? atoi(tune_env("DRAM_EW_SHAPE")) : 2;   // 0 capped, 1 one-shot, 2 + nt
  return v;
}
// streaming cache policy only for tensors that cannot stay in the 256 MiB Infinity Cache anyway: on a 67 MB tensor
// (producer -> consumer inside the cache) the non-temporal forms cost 5-15 %, on a 537 MB one they gain 7-10 %
static inline bool ew_stream(long long tensor_bytes) { return ew_shape() >= 2 && tensor_bytes >= (256LL << 20); }
static inline int ew_blocks(long items, int per_block, int cap) {
  long b = (items + per_block - 1) / per_block;
  if (b < 1) b = 1;
  if (ew_shape() == 0 && b > cap) b = cap;
  return (int)(b > 0x7fffffffL ? 0x7fffffffL : b);
}

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// XCD-aware bijective remap of the linear workgroup id (8 XCDs, round-robin
// dispatch): each XCD receives a contiguous run of logical tiles so that
// neighbouring tiles (which share halo rows / weight panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Trilinear source index -- PyTorch area_pixel_compute_source_index with align_corners=True:
//   scale = (in-1)/(out-1) (float; 0 when out == 1), src = scale*dst, i0 = (int)src,
//   i1 = i0 + (i0 < in-1), w1 = src - i0, w0 = 1 - w1.   (pool_up.hip, upmix.hip)
__device__ __forceinline__ void lin_src(int dst, float scale, int in, int& i0, int& i1, float& w0, float& w1) {
  const float s = scale * (float)dst;
  i0 = (int)s;
  if (i0 > in - 1) i0 = in - 1;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  w1 = s - (float)i0;
  w0 = 1.f - w1;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
