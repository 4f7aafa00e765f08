// optim.hip -- multi-tensor Adam / SGD weight update (one launch for every parameter).
// Replaces torch.optim.Adam.step at reference models.py:385-387 / :689-691 (defaults
// betas=(0.9,0.999), eps=1e-8, weight_decay=0, no amsgrad) and the SGD(momentum,
// weight_decay) alternative the reference keeps as arguments (train.py:25,27;
// models.py:388-389).  HBM-bound: 28 B/param (read p,g,m,v; write p,m,v).
#include "common.h"

namespace {

// ONE element's update, the same instruction sequence on the vector and the scalar path (contraction pinned: left to
// the compiler the two paths fused differently, and a parameter whose gradient arrived 4-byte aligned -- a view of the
// data-parallel step's small-gradient bucket -- moved one ulp away from the same update of an aligned gradient).
__device__ __forceinline__ void adam1(float& p, const float g, float& m, float& v, const float b1, const float b2,
                                      const float eps, const float wd, const float step, const float rs2,
                                      const float gscale) {
#pragma clang fp contract(off)
  float gg = g * gscale;
  if (wd != 0.f) gg = __builtin_fmaf(wd, p, gg);
  m = __builtin_fmaf(b1, m, (1.f - b1) * gg);
  v = __builtin_fmaf(b2, v, ((1.f - b2) * gg) * gg);
  p -= (step * m) / __builtin_fmaf(sqrtf(v), rs2, eps);
}

__device__ __forceinline__ void adam_chunk(const DramTensorRef* __restrict__ table,
                                           const DramChunkRef* __restrict__ chunks, float lr, float b1, float b2,
                                           float eps, float wd, float bc1, float bc2, float gscale) {
  const DramChunkRef ch = chunks[blockIdx.x];
  const DramTensorRef t = table[ch.tensor];
  const long n = t.n - ch.offset < DRAM_OPT_CHUNK ? t.n - ch.offset : DRAM_OPT_CHUNK;
  float* p = t.p + ch.offset;
  const float* g = t.g + ch.offset;
  float* m = t.m + ch.offset;
  float* v = t.v + ch.offset;
  const float step = lr / bc1;
  const float rs2 = 1.f / sqrtf(bc2);
  // torch: denom = sqrt(v)/sqrt(bc2) + eps ; p -= (lr/bc1) * m / denom
  const bool vec = ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0);
  if (vec) {
    const long n4 = n >> 2;
    for (long i = threadIdx.x; i < n4; i += 256) {
      float4 pv = reinterpret_cast<float4*>(p)[i];
      const float4 gv = reinterpret_cast<const float4*>(g)[i];
      float4 mv = reinterpret_cast<float4*>(m)[i];
      float4 vv = reinterpret_cast<float4*>(v)[i];
      adam1(pv.x, gv.x, mv.x, vv.x, b1, b2, eps, wd, step, rs2, gscale);
      adam1(pv.y, gv.y, mv.y, vv.y, b1, b2, eps, wd, step, rs2, gscale);
      adam1(pv.z, gv.z, mv.z, vv.z, b1, b2, eps, wd, step, rs2, gscale);
      adam1(pv.w, gv.w, mv.w, vv.w, b1, b2, eps, wd, step, rs2, gscale);
      reinterpret_cast<float4*>(p)[i] = pv;
      reinterpret_cast<float4*>(m)[i] = mv;
      reinterpret_cast<float4*>(v)[i] = vv;
    }
    for (long i = (n4 << 2) + threadIdx.x; i < n; i += 256) adam1(p[i], g[i], m[i], v[i], b1, b2, eps, wd, step, rs2, gscale);
  } else {
    // (16-byte alignment is not guaranteed: a data-parallel step hands over small gradients as views of one bucket)
    for (long i = threadIdx.x; i < n; i += 256) adam1(p[i], g[i], m[i], v[i], b1, b2, eps, wd, step, rs2, gscale);
  }
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const DramTensorRef* __restrict__ table,
                                                         const DramChunkRef* __restrict__ chunks, float lr, float b1,
                                                         float b2, float eps, float wd, float bc1, float bc2,
                                                         float gscale) {
  adam_chunk(table, chunks, lr, b1, b2, eps, wd, bc1, bc2, gscale);
}

// Graph-replayable form: every hyper-parameter AND the step count live in device memory, so a captured
// hipGraph of the train step stays valid while lr decays (ExponentialLR) and the bias corrections change.
// hyper = [lr, b1, b2, eps, wd, grad_scale, step]; step is advanced by adam_advance_kernel first.  The step slot
// holds the BITS of an int32 (exact for 2^31 steps; a float counter would stall at 2^24).
__global__ void adam_advance_kernel(float* __restrict__ hyper) {
  if (threadIdx.x == 0 && blockIdx.x == 0) reinterpret_cast<int*>(hyper)[6] += 1;
}
__global__ __launch_bounds__(256) void adam_multi_dev_kernel(const DramTensorRef* __restrict__ table,
                                                             const DramChunkRef* __restrict__ chunks,
                                                             const float* __restrict__ hyper) {
  const float b1 = hyper[1], b2 = hyper[2];
  const double t = (double)reinterpret_cast<const int*>(hyper)[6];
  // torch: bias_correction = 1 - beta ** step, evaluated in double on the host
  const float bc1 = (float)(1.0 - pow((double)b1, t)), bc2 = (float)(1.0 - pow((double)b2, t));
  adam_chunk(table, chunks, hyper[0], b1, b2, hyper[3], hyper[4], bc1, bc2, hyper[5]);
}

__global__ __launch_bounds__(256) void sgd_multi_kernel(const DramTensorRef* __restrict__ table,
                                                        const DramChunkRef* __restrict__ chunks, float lr, float mom,
                                                        float wd, int first, float gscale) {
  const DramChunkRef ch = chunks[blockIdx.x];
  const DramTensorRef t = table[ch.tensor];
  const long n = t.n - ch.offset < DRAM_OPT_CHUNK ? t.n - ch.offset : DRAM_OPT_CHUNK;
  float* p = t.p + ch.offset;
  const float* g = t.g + ch.offset;
  float* m = t.m ? t.m + ch.offset : nullptr;
  for (long i = threadIdx.x; i < n; i += 256) {
    float gg = g[i] * gscale;
    if (wd != 0.f) gg += wd * p[i];
    if (mom != 0.f && m) {
      const float bb = first ? gg : mom * m[i] + gg;
      m[i] = bb;
      gg = bb;
    }
    p[i] -= lr * gg;
  }
}

}  // namespace

extern "C" int dram_adam_multi(const DramTensorRef* table, const DramChunkRef* chunks, int nchunks, float lr,
                               float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                               float bias_corr2, float grad_scale, dram_stream_t stream) {
  if (!table || !chunks || nchunks < 1 || bias_corr1 <= 0.f || bias_corr2 <= 0.f) return DRAM_ERR_BAD_ARG;
  // 28 B per parameter (r p,g,m,v; w p,m,v); the chunk count bounds the parameter count from above
  DramProf prof(DRAM_FAM_OPTIM, 0, 0.0, 28.0 * (double)nchunks * 16384.0, (hipStream_t)stream);
  hipLaunchKernelGGL(adam_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunks, lr, beta1,
                     beta2, eps, weight_decay, bias_corr1, bias_corr2, grad_scale);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_adam_multi_dev(const DramTensorRef* table, const DramChunkRef* chunks, int nchunks, float* hyper,
                                   dram_stream_t stream) {
  if (!table || !chunks || nchunks < 1 || !hyper) return DRAM_ERR_BAD_ARG;
  DramProf prof(DRAM_FAM_OPTIM, 2, 0.0, 28.0 * (double)nchunks * 16384.0, (hipStream_t)stream);
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, hyper);
  DRAM_LAUNCH_CHECK();
  hipLaunchKernelGGL(adam_multi_dev_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunks,
                     (const float*)hyper);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_sgd_multi(const DramTensorRef* table, const DramChunkRef* chunks, int nchunks, float lr,
                              float momentum, float weight_decay, int first_step, float grad_scale,
                              dram_stream_t stream) {
  if (!table || !chunks || nchunks < 1) return DRAM_ERR_BAD_ARG;
  DramProf prof(DRAM_FAM_OPTIM, 1, 0.0, (momentum != 0.f ? 20.0 : 12.0) * (double)nchunks * 16384.0, (hipStream_t)stream);
  hipLaunchKernelGGL(sgd_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunks, lr, momentum,
                     weight_decay, first_step, grad_scale);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_version(void) { return DRAM_ABI_VERSION; }
extern "C" const char* dram_build_info(void) { return "libdram_hip gfx950 fp32-mfma"; }
#ifndef DRAM_ABI_HASH
#error "build through _build.py (it passes -DDRAM_ABI_HASH=<sha1 of include/dram_hip.h>)"
#endif
extern "C" const char* dram_abi_hash(void) { return DRAM_ABI_HASH; }

extern "C" unsigned long long dram_stream_capture_id(dram_stream_t stream) {
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  if (hipStreamGetCaptureInfo((hipStream_t)stream, &status, &id) != hipSuccess) return 0;
  if (status != hipStreamCaptureStatusActive) return 0;
  return id ? id : ~0ull;                            // (never 0 while capturing)
}
