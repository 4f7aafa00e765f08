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

// dword-per-lane forms (the Winograd transforms move one channel per lane: 256 B per wave instruction)
template <int U>
__global__ __launch_bounds__(256) void kD(const float* __restrict__ x, float* __restrict__ y, long n) {   // n floats
  const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
  float v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = x[base + u * 256];
#pragma unroll
  for (int u = 0; u < U; ++u) if (base + u * 256 < n) y[base + u * 256] = fmaxf(v[u] * 2.f + 1.f, 0.f);
}
// scatter of the Winograd input transform: a wave reads P consecutive 256-B rows and writes them to P planes that lie
// `plane` floats apart (one 256-B piece per plane), consecutive waves write consecutive pieces of every plane
template <int P>
__global__ __launch_bounds__(256) void kS(const float* __restrict__ x, float* __restrict__ y, long nw, long plane) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= nw) return;
  float v[P];
#pragma unroll
  for (int p = 0; p < P; ++p) v[p] = x[(w * P + p) * 64 + lane];
#pragma unroll
  for (int p = 0; p < P; ++p) y[p * plane + w * 64 + lane] = v[p] * 2.f + 1.f;
}

// F: one-shot blocks that each sweep K consecutive 4-KB sub-chunks (a kernel that needs per-block partial results and
// cannot afford one partial row per 4 KB): 3 streams (two reads, one write) like bn_bwd_apply
template <int K>
__global__ __launch_bounds__(256) void kF(const float4* __restrict__ x, const float4* __restrict__ x2, float4* __restrict__ y, long n) {
  const long base = (long)blockIdx.x * (256 * 2 * K) + threadIdx.x;
  float acc = 0.f;
#pragma unroll 1
  for (int k = 0; k < K; ++k) {
    const long i0 = base + k * 512;
    float4 a[2], b[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) if (i0 + u * 256 < n) { a[u] = x[i0 + u * 256]; b[u] = x2[i0 + u * 256]; }
#pragma unroll
    for (int u = 0; u < 2; ++u) if (i0 + u * 256 < n) { float4 o = a[u]; o.x = fmaxf(o.x * b[u].x + 1.f, 0.f); acc += o.x; y[i0 + u * 256] = o; }
  }
  if (acc == 123.456f) y[0].y = acc;
}

// write-heavy scatter (the input transform reads 64 voxels and writes 216 points per tile: 1 : 3.4): a wave reads R rows
// and writes P planes
template <int P, int R>
__global__ __launch_bounds__(256) void kW(const float* __restrict__ x, float* __restrict__ y, long nw, long plane) {
  const long w = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (w >= nw) return;
  float v[R];
#pragma unroll
  for (int p = 0; p < R; ++p) v[p] = x[(w * R + p) * 64 + lane];
#pragma unroll
  for (int p = 0; p < P; ++p) y[p * plane + w * 64 + lane] = v[p % R] * 2.f + (float)p;
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
    {
      float4* x2; CK(hipMalloc(&x2, n * 16)); CK(hipMemset(x2, 0x3c, n * 16));
      auto rep3 = [&](const char* name, float ms) { printf("%5ld MiB  %-28s %8.1f us  %7.0f GB/s (3 streams)\n", mb, name, ms * 1e3, 3.0 * n * 16 / ms / 1e6); fflush(stdout); };
#define F_(K) rep3("F 2 reads + 1 write, K=" #K, timeit([&] { hipLaunchKernelGGL(kF<K>, dim3((unsigned)((n + 512 * K - 1) / (512 * K))), dim3(256), 0, 0, x, (const float4*)x2, y, n); }))
      F_(1); F_(2); F_(4); F_(8); F_(16); F_(32);
      CK(hipFree(x2));
    }
    {
      const long nf = n * 4;
      rep("D dword one-shot U=1", timeit([&] { hipLaunchKernelGGL(kD<1>, dim3((unsigned)((nf + 255) / 256)), dim3(256), 0, 0, (const float*)x, (float*)y, nf); }));
      rep("D dword one-shot U=4", timeit([&] { hipLaunchKernelGGL(kD<4>, dim3((unsigned)((nf + 1023) / 1024)), dim3(256), 0, 0, (const float*)x, (float*)y, nf); }));
      rep("D dword one-shot U=16", timeit([&] { hipLaunchKernelGGL(kD<16>, dim3((unsigned)((nf + 4095) / 4096)), dim3(256), 0, 0, (const float*)x, (float*)y, nf); }));
      const long nw27 = nf / 64 / 27, nw108 = nf / 64 / 108;
      rep("S 27 planes per wave", timeit([&] { hipLaunchKernelGGL(kS<27>, dim3((unsigned)((nw27 + 3) / 4)), dim3(256), 0, 0, (const float*)x, (float*)y, nw27, nw27 * 64); }));
      {
        const long nwW = nf / 64 / 108;
        auto repw = [&](const char* name, float ms, double bytes) { printf("%5ld MiB  %-28s %8.1f us  %7.0f GB/s\n", mb, name, ms * 1e3, bytes / ms / 1e6); fflush(stdout); };
        repw("W 32 rows -> 108 planes", timeit([&] { hipLaunchKernelGGL((kW<108, 32>), dim3((unsigned)((nwW + 3) / 4)), dim3(256), 0, 0, (const float*)x, (float*)y, nwW, nwW * 64); }), (double)nwW * 256 * 140);
        repw("W 1 row -> 108 planes (write only)", timeit([&] { hipLaunchKernelGGL((kW<108, 1>), dim3((unsigned)((nwW + 3) / 4)), dim3(256), 0, 0, (const float*)x, (float*)y, nwW, nwW * 64); }), (double)nwW * 256 * 109);
      }
      rep("S 108 planes per wave", timeit([&] { hipLaunchKernelGGL(kS<108>, dim3((unsigned)((nw108 + 3) / 4)), dim3(256), 0, 0, (const float*)x, (float*)y, nw108, nw108 * 64); }));
    }
    CK(hipFree(x)); CK(hipFree(y));
  }
  return 0;
}
