// common.h -- shared helpers for the gfx950 kernels of libdram_hip.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
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

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// XCD-aware bijective remap of the linear workgroup id (8 XCDs, round-robin
// dispatch): each XCD receives a contiguous run of logical tiles so that
// neighbouring tiles (which share halo rows / weight panels) hit the same L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
