// stream_probe.hip -- which launch shape streams y = f(x) fastest on gfx950?  (hipcc --offload-arch=gfx950 -O3)
//   A: grid-stride loop, 8192 blocks, U loads in flight, accesses `stride` apart      (bn_apply_fast_kernel's shape)
//   B: one-shot blocks, each block owns a contiguous run of U * 4 KB                  (torch's vectorized kernel)
//   C: B with non-temporal stores;  D: B with non-temporal loads and stores
//   E: grid-stride over contiguous U * 4 KB runs (persistent blocks, run = consecutive)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int U>
__global__ __launch_bounds__(256) void kA(const float4* __restrict__ x, float4* __restrict__ y, long n) {
  const long stride = (long)gridDim.x * 256;
  for (long i0 = blockIdx.x * 256L + threadIdx.x; i0 < n; i0 += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * stride < n) v[u] = x[i0 + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) if (i0 + u * stride < n) { float4 o = v[u]; o.x = fmaxf(o.x * 2.f + 1.f, 0.f); y[i0 + u * stride] = o; }
  }
}
template <int U, int NT>
__global__ __launch_bounds__(256) void kB(const float4* __restrict__ x, float4* __restrict__ y, long n) {
  const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
  float4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) { if (NT >= 2) { f4v t = __builtin_nontemporal_load((const f4v*)(x + base + u * 256)); v[u] = make_float4(t.x, t.y, t.z, t.w); } else v[u] = x[base + u * 256]; }
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) {
    float4 o = v[u]; o.x = fmaxf(o.x * 2.f + 1.f, 0.f);
    if (NT >= 1) { f4v t = {o.x, o.y, o.z, o.w}; __builtin_nontemporal_store(t, (f4v*)(y + base + u * 256)); } else y[base + u * 256] = o;
  }
}
template <int U>
__global__ __launch_bounds__(256) void kE(const float4* __restrict__ x, float4* __restrict__ y, long n) {
  for (long base = (long)blockIdx.x * (256 * U) + threadIdx.x; base < n; base += (long)gridDim.x * 256 * U) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = x[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { float4 o = v[u]; o.x = fmaxf(o.x * 2.f + 1.f, 0.f); y[base + u * 256] = o; }
  }
}

template <typename F> float timeit(F f) {
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(a));
  for (int i = 0; i < 20; ++i) f();
  CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
  float ms; CK(hipEventElapsedTime(&ms, a, b)); return ms / 20;
}
int main() {
  for (long mb : {64L, 512L, 1024L}) {
    const long n = mb * (1 << 20) / 16;
    float4 *x, *y; CK(hipMalloc(&x, n * 16)); CK(hipMalloc(&y, n * 16)); CK(hipMemset(x, 0x3c, n * 16));
    auto rep = [&](const char* name, float ms) { printf("%5ld MiB  %-28s %8.1f us  %7.0f GB/s\n", mb, name, ms * 1e3, 2.0 * n * 16 / ms / 1e6); fflush(stdout); };
    for (int g : {2048, 4096, 8192, 16384}) {
      char nm[64];
      snprintf(nm, 64, "A grid-stride U=4 g=%d", g); rep(nm, timeit([&] { hipLaunchKernelGGL(kA<4>, dim3(g), dim3(256), 0, 0, x, y, n); }));
      snprintf(nm, 64, "A grid-stride U=1 g=%d", g); rep(nm, timeit([&] { hipLaunchKernelGGL(kA<1>, dim3(g), dim3(256), 0, 0, x, y, n); }));
      snprintf(nm, 64, "E persistent runs U=4 g=%d", g); rep(nm, timeit([&] { hipLaunchKernelGGL(kE<4>, dim3(g), dim3(256), 0, 0, x, y, n); }));
    }
#define B_(U, NT) rep("B one-shot U=" #U " nt=" #NT, timeit([&] { hipLaunchKernelGGL((kB<U, NT>), dim3((unsigned)((n + 256 * U - 1) / (256 * U))), dim3(256), 0, 0, x, y, n); }))
    B_(1, 0); B_(2, 0); B_(4, 0); B_(8, 0); B_(4, 1); B_(4, 2); B_(8, 1);
    CK(hipFree(x)); CK(hipFree(y));
  }
  return 0;
}
