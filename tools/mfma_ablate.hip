// Ablation ladder for the fp32-MFMA implicit-GEMM loop structure (standalone, no torch):
//   level 0: 64 MFMAs per iteration only          level 1: + ds_read_b128 operand reads
//   level 2: + two barriers per iteration          level 3: + 10 ds_write_b128 per thread
//   level 4: + 10 global_load_dwordx4 per thread (L2-resident source), waited before the writes
//   level 5: levels 0-2 + 10 global_load_lds_dwordx4 per wave (LDS-DMA, no VGPR staging, no ds_write)
//   level 6: level 5 with double-buffered LDS and ONE barrier per iteration (the v2 kernel's loop)
//   level 7: level 4 with the loads issued AFTER the MFMAs (and waited at the next ds_write)
// hipcc -O3 --offload-arch=gfx950 tools/mfma_ablate.hip -o /tmp/mfma_ablate && /tmp/mfma_ablate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int LDK = 36;

template <int LEVEL>
__global__ __launch_bounds__(256, 2) void k(const float* __restrict__ src, float* out, int iters) {
  __shared__ __attribute__((aligned(1024))) float lds[(LEVEL == 6 ? 2 : 1) * (256 + 64) * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < (256 + 64) * LDK; i += 256) lds[i] = (float)(i % 7) * 0.01f;
  __syncthreads();
  f32x16 acc[2][2];
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const float* a_rd = &lds[(wave * 64 + li) * LDK + 4 * lh];
  const float* b_rd = &lds[256 * LDK + li * LDK + 4 * lh];
  float4 r[10];
  for (int p = 0; p < 10; ++p) r[p] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float* gp = src + (blockIdx.x % 64) * 81920 + tid * 4;
  f32x4 fa0 = {1.f, 2.f, 3.f, 4.f}, fa1 = fa0, fb0 = fa0, fb1 = fa0;
  for (int it = 0; it < iters; ++it) {
    if (LEVEL >= 2) __syncthreads();
    if (LEVEL == 5 || LEVEL == 6) {
      float* dst = lds + ((LEVEL == 6) ? ((it + 1) & 1) * (256 + 64) * LDK : 0) + wave * 2560;
#pragma unroll
      for (int p = 0; p < 10; ++p)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(gp + ((it * 10 + p) % 20) * 1024),
                                         (__attribute__((address_space(3))) void*)(dst + p * 256), 16, 0, 0);
    }
    if (LEVEL == 3 || LEVEL == 4 || LEVEL == 7) {
#pragma unroll
      for (int p = 0; p < 10; ++p) *reinterpret_cast<float4*>(&lds[((tid >> 3) + 32 * p) * LDK + (tid & 7) * 4]) = r[p];
    }
    if (LEVEL >= 2 && LEVEL != 6) __syncthreads();
    if (LEVEL == 4) {
#pragma unroll
      for (int p = 0; p < 10; ++p) r[p] = *reinterpret_cast<const float4*>(gp + ((it * 10 + p) % 20) * 1024);
    }
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      if (LEVEL >= 1) {
        fa0 = *reinterpret_cast<const f32x4*>(a_rd + gk * 8);
        fa1 = *reinterpret_cast<const f32x4*>(a_rd + 32 * LDK + gk * 8);
        fb0 = *reinterpret_cast<const f32x4*>(b_rd + gk * 8);
        fb1 = *reinterpret_cast<const f32x4*>(b_rd + 32 * LDK + gk * 8);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb0[e], acc[0][0], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb0[e], acc[1][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa0[e], fb1[e], acc[0][1], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa1[e], fb1[e], acc[1][1], 0, 0, 0);
      }
    }
    if (LEVEL == 7) {
#pragma unroll
      for (int p = 0; p < 10; ++p) r[p] = *reinterpret_cast<const float4*>(gp + ((it * 10 + p) % 20) * 1024);
    }
  }
  float s = 0.f;
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) for (int e = 0; e < 16; ++e) s += acc[a][b][e];
  for (int p = 0; p < 10; ++p) s += r[p].x;
  out[blockIdx.x * 256 + tid] = s;
}

template <int LEVEL>
void run(const float* src, float* out, int grid, int iters, size_t dyn) {
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL((k<LEVEL>), dim3(grid), dim3(256), dyn, 0, src, out, iters);
  hipDeviceSynchronize();
  hipEventRecord(a);
  hipLaunchKernelGGL((k<LEVEL>), dim3(grid), dim3(256), dyn, 0, src, out, iters);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double flops = (double)grid * 4 * 64.0 * iters * 4096.0;
  printf("level %d grid %5d dynLDS %6zu: %8.3f ms  %7.1f TFLOP/s\n", LEVEL, grid, dyn, ms, flops / ms / 1e9);
}

int main() {
  float *src, *out;
  hipMalloc(&src, 64 * 81920 * 4 + 4 * 1024 * 100);
  hipMemset(src, 0, 64 * 81920 * 4 + 4 * 1024 * 100);
  hipMalloc(&out, 8192 * 256 * 4);
  for (size_t dyn : {(size_t)0, (size_t)9000, (size_t)40000}) {   // 3, 2, 1 workgroups per CU
    for (int grid : {8192}) {
      run<0>(src, out, grid, 108, dyn); run<1>(src, out, grid, 108, dyn); run<2>(src, out, grid, 108, dyn);
      run<3>(src, out, grid, 108, dyn); run<4>(src, out, grid, 108, dyn); run<5>(src, out, grid, 108, dyn);
      run<6>(src, out, grid, 108, dyn); run<7>(src, out, grid, 108, dyn);
    }
  }
  return 0;
}
