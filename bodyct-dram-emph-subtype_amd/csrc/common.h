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
