// MFMA throughput vs the number of independent accumulator chains per wave (2 waves per SIMD):
//   v_mfma_f32_32x32x2_f32 with 1 / 2 / 4 chains, v_mfma_f32_16x16x4_f32 with 4 / 8 chains.
// hipcc -O3 --offload-arch=gfx950 tools/mfma_chain.hip -o /tmp/mfma_chain && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
  f32x16 acc[CH];
  for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int CH>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
  f32x4 acc[CH];
  for (int c = 0; c < CH; ++c) for (int e = 0; e < 4; ++e) acc[c][e] = 0.f;
  float a = threadIdx.x * 0.001f, b = 1.f + threadIdx.x * 0.002f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 32 / CH; ++r)
#pragma unroll
      for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
  }
  float s = 0.f;
  for (int c = 0; c < CH; ++c) for (int e = 0; e < 4; ++e) s += acc[c][e];
  out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <typename F>
void run(const char* name, F launch, double flops_per_iter_per_wave) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  launch(100);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double tf = flops_per_iter_per_wave * iters * 8.0 * 256.0 / (ms * 1e-3) / 1e12;
  printf("%-28s %8.3f ms  %7.1f TFLOP/s\n", name, ms, tf);
}

int main() {
  float* out; hipMalloc(&out, 256 * 512 * 4);
  run("32x32x2, 1 chain/wave", [&](int it) { hipLaunchKernelGGL(k32<1>, dim3(256), dim3(512), 0, 0, out, it); }, 16 * 4096.0);
  run("32x32x2, 2 chains/wave", [&](int it) { hipLaunchKernelGGL(k32<2>, dim3(256), dim3(512), 0, 0, out, it); }, 16 * 4096.0);
  run("32x32x2, 4 chains/wave", [&](int it) { hipLaunchKernelGGL(k32<4>, dim3(256), dim3(512), 0, 0, out, it); }, 16 * 4096.0);
  run("16x16x4, 2 chains/wave", [&](int it) { hipLaunchKernelGGL(k16<2>, dim3(256), dim3(512), 0, 0, out, it); }, 32 * 2048.0);
  run("16x16x4, 4 chains/wave", [&](int it) { hipLaunchKernelGGL(k16<4>, dim3(256), dim3(512), 0, 0, out, it); }, 32 * 2048.0);
  run("16x16x4, 8 chains/wave", [&](int it) { hipLaunchKernelGGL(k16<8>, dim3(256), dim3(512), 0, 0, out, it); }, 32 * 2048.0);
  return 0;
}
