// head_loss.hip -- 1x1x1 heads + pooled scores, and the fused dRAM segmentation losses.
// Replaces: fcs Conv3d(32,n,1) + adaptive_avg_pool3d (reference med3d.py:283-284),
// sigmoid(fcs) + nearest-resized-lung masked mean (med3d.py:382-387), and the ~25
// elementwise/reduction torch ops of _segmentation_loss (models.py:523-531, metrics.py:10-37,
// label prep models.py:567-570), each with its backward.  One HBM pass per kernel.
#include "common.h"

namespace {

// F.interpolate(mode='nearest') source index: min(floor(dst * in/out), in-1)
__device__ __forceinline__ int nearest_src(int dst, float scale, int in) {
  const int s = (int)floorf((float)dst * scale);
  return s < in - 1 ? s : in - 1;
}

struct NearGeom {
  int Dl, Hl, Wl;
  float sz, sy, sx;
};

__device__ __forceinline__ long near_index(const NearGeom& n, long b, int z, int y, int x) {
  return ((b * n.Dl + nearest_src(z, n.sz, n.Dl)) * n.Hl + nearest_src(y, n.sy, n.Hl)) * (long)n.Wl +
         nearest_src(x, n.sx, n.Wl);
}

// ----------------------------------------------------------------------------- head fwd
template <int NOT, typename T>
__global__ __launch_bounds__(256) void head_fwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ bias, const float* __restrict__ lungs,
                                                       NearGeom ng, float* __restrict__ dense,
                                                       float* __restrict__ partial, int D, int H, int W, int NO,
                                                       int sigmoid, int nblk) {
  __shared__ float wl[NOT * 32 + NOT];
  __shared__ float red[4][NOT + 1];
  const int tid = threadIdx.x;
  for (int i = tid; i < NO * 32; i += 256) wl[i] = w[i];
  for (int i = tid; i < NO; i += 256) wl[NOT * 32 + i] = bias[i];
  __syncthreads();
  const long b = blockIdx.y;
  const long vps = (long)D * H * W;
  float acc[NOT + 1];
#pragma unroll
  for (int c = 0; c <= NOT; ++c) acc[c] = 0.f;
  for (long v = blockIdx.x * 256L + tid; v < vps; v += (long)gridDim.x * 256L) {
    // the weights are re-read from LDS for every voxel: left alone the compiler hoists all 32 NOT reads out of this
    // loop -- 366 VGPRs for the 9-output head, one wave per SIMD under a kernel that lives on loads in flight
    asm volatile("" ::: "memory");
    float4 xv[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = ld4<T>(x, (b * vps + v) * 32 + 4 * k);
    float L = 1.f;
    if (lungs) {
      long r = v;
      const int xo = (int)(r % W); r /= W;
      const int yo = (int)(r % H);
      const int zo = (int)(r / H);
      L = lungs[near_index(ng, b, zo, yo, xo)];
    }
#pragma unroll
    for (int c = 0; c < NOT; ++c) {
      if (c < NO) {
        float s = wl[NOT * 32 + c];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          s += xv[k].x * wl[c * 32 + 4 * k] + xv[k].y * wl[c * 32 + 4 * k + 1] + xv[k].z * wl[c * 32 + 4 * k + 2] +
               xv[k].w * wl[c * 32 + 4 * k + 3];
        }
        if (sigmoid) s = 1.f / (1.f + expf(-s));
        dense[(b * NO + c) * vps + v] = s;
        acc[c] += sigmoid ? s * L : s;
      }
    }
    acc[NOT] += L;
  }
#pragma unroll
  for (int c = 0; c <= NOT; ++c) {
    const float s = wave_sum(acc[c]);
    if ((tid & 63) == 0) red[tid >> 6][c] = s;
  }
  __syncthreads();
  if (tid <= NO) {
    const int c = tid < NO ? tid : NOT;
    partial[((long)b * nblk + blockIdx.x) * (NO + 1) + tid] = red[0][c] + red[1][c] + red[2][c] + red[3][c];
  }
}

// ----------------------------------------------------------------------------- head bwd
template <int NOT, typename T>
__global__ __launch_bounds__(256) void head_bwd_kernel(const T* __restrict__ x, const float* __restrict__ w,
                                                       const float* __restrict__ dense, const float* __restrict__ gdense,
                                                       const float* __restrict__ gpool, const float* __restrict__ lungs,
                                                       NearGeom ng, T* __restrict__ dx, float* __restrict__ wpartial,
                                                       int D, int H, int W, int NO, int sigmoid, int nblk) {
  constexpr int SLOTS = (NOT * 33 + 255) / 256;
  __shared__ float wl[NOT * 32];
  __shared__ float xs[256 * 33];
  __shared__ float ds[256 * NOT];
  const int tid = threadIdx.x;
  for (int i = tid; i < NO * 32; i += 256) wl[i] = w[i];
  const long b = blockIdx.y;
  const long vps = (long)D * H * W;
  float gp[NOT];
#pragma unroll
  for (int c = 0; c < NOT; ++c) gp[c] = (c < NO) ? gpool[b * NO + c] : 0.f;
  float wacc[SLOTS];
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) wacc[s] = 0.f;
  __syncthreads();
  const long ntile = (vps + 255) / 256;
  for (long t = blockIdx.x; t < ntile; t += gridDim.x) {
    const long v = t * 256 + tid;
    const bool ok = v < vps;
    float4 xv[8];
    float dp[NOT];
#pragma unroll
    for (int c = 0; c < NOT; ++c) dp[c] = 0.f;
#pragma unroll
    for (int k = 0; k < 8; ++k) xv[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ok) {
#pragma unroll
      for (int k = 0; k < 8; ++k) xv[k] = ld4<T>(x, (b * vps + v) * 32 + 4 * k);
      float L = 1.f;
      if (lungs) {
        long r = v;
        const int xo = (int)(r % W); r /= W;
        const int yo = (int)(r % H);
        const int zo = (int)(r / H);
        L = lungs[near_index(ng, b, zo, yo, xo)];
      }
#pragma unroll
      for (int c = 0; c < NOT; ++c) {
        if (c < NO) {
          float g = gp[c] * (sigmoid ? L : 1.f);
          if (gdense) g += gdense[(b * NO + c) * vps + v];
          if (sigmoid) {
            const float s = dense[(b * NO + c) * vps + v];
            g *= s * (1.f - s);
          }
          dp[c] = g;
        }
      }
      float4 o[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) o[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
      for (int c = 0; c < NOT; ++c) {
        if (c < NO) {
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            o[k].x += dp[c] * wl[c * 32 + 4 * k];
            o[k].y += dp[c] * wl[c * 32 + 4 * k + 1];
            o[k].z += dp[c] * wl[c * 32 + 4 * k + 2];
            o[k].w += dp[c] * wl[c * 32 + 4 * k + 3];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) st4<T>(dx, (b * vps + v) * 32 + 4 * k, o[k]);
    }
    __syncthreads();  // previous tile's LDS consumers are done
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      xs[tid * 33 + 4 * k] = xv[k].x; xs[tid * 33 + 4 * k + 1] = xv[k].y;
      xs[tid * 33 + 4 * k + 2] = xv[k].z; xs[tid * 33 + 4 * k + 3] = xv[k].w;
    }
    xs[tid * 33 + 32] = ok ? 1.f : 0.f;
#pragma unroll
    for (int c = 0; c < NOT; ++c) ds[tid * NOT + c] = dp[c];
    __syncthreads();
#pragma unroll
    for (int s = 0; s < SLOTS; ++s) {
      const int idx = s * 256 + tid;
      if (idx < NO * 33) {
        const int c = idx / 33, k = idx - c * 33;
        float a = 0.f;
        for (int vv = 0; vv < 256; ++vv) a += ds[vv * NOT + c] * xs[vv * 33 + k];
        wacc[s] += a;
      }
    }
  }
#pragma unroll
  for (int s = 0; s < SLOTS; ++s) {
    const int idx = s * 256 + tid;
    if (idx < NO * 33) wpartial[((long)b * nblk + blockIdx.x) * (NO * 33) + idx] = wacc[s];
  }
}

// ----------------------------------------------------------------------------- seg loss
__global__ __launch_bounds__(256) void segloss_fwd_kernel(const float* __restrict__ cle, const float* __restrict__ pse,
                                                          const float* __restrict__ lungs, const float* __restrict__ ems,
                                                          const float* __restrict__ binary, NearGeom ng,
                                                          float* __restrict__ partial, int B, int D, int H, int W,
                                                          float smoothness) {
  __shared__ float red[4][6];
  const long vps = (long)D * H * W;
  const long total = vps * B;
  float s[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long b = i / vps;
    long r = i - b * vps;
    const int xo = (int)(r % W); r /= W;
    const int yo = (int)(r % H);
    const int zo = (int)(r / H);
    const long li = near_index(ng, b, zo, yo, xo);
    const float L = lungs[li];
    const float t = ems[li] * binary[b];
    const float c = cle[i], p_ = pse[i];
    float p = c + p_;
    p = fminf(fmaxf(p, 0.f), 1.f);
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float ptc = fminf(fmaxf(pt, 1e-6f), 1.f - 1e-6f);
    const float cw = smoothness * L + (1.f - L);
    const float nl = -cw * logf(ptc);
    s[0] += t;
    s[1] += nl * t;
    s[2] += nl * (1.f - t);
    s[3] += (c * L) * (p_ * L);
    s[4] += c * L;
    s[5] += p_ * L;
  }
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const float v = wave_sum(s[k]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 6)
    partial[(long)blockIdx.x * 6 + threadIdx.x] =
        red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

// coef: [0] g_mul*2/den  [1] g_mul*(2I+eps)/den^2  [2] g_seg*alpha/sum_w  [3] g_seg*(1-alpha)/sum_w
__global__ void segloss_bwd_kernel(const float* __restrict__ cle, const float* __restrict__ pse,
                                   const float* __restrict__ lungs, const float* __restrict__ ems,
                                   const float* __restrict__ binary, NearGeom ng, const float* __restrict__ coef,
                                   float* __restrict__ gcle, float* __restrict__ gpse, int B, int D, int H, int W,
                                   float smoothness) {
  const long vps = (long)D * H * W;
  const long total = vps * B;
  const float k0 = coef[0], k1 = coef[1], k2 = coef[2], k3 = coef[3];
  for (long i = blockIdx.x * 256L + threadIdx.x; i < total; i += (long)gridDim.x * 256L) {
    const long b = i / vps;
    long r = i - b * vps;
    const int xo = (int)(r % W); r /= W;
    const int yo = (int)(r % H);
    const int zo = (int)(r / H);
    const long li = near_index(ng, b, zo, yo, xo);
    const float L = lungs[li];
    const float t = ems[li] * binary[b];
    const float c = cle[i], p_ = pse[i];
    const float sum = c + p_;
    const float p = fminf(fmaxf(sum, 0.f), 1.f);
    const float pt = p * t + (1.f - p) * (1.f - t);
    const float ptc = fminf(fmaxf(pt, 1e-6f), 1.f - 1e-6f);
    const float cw = smoothness * L + (1.f - L);
    // d/dp of -cw*log(ptc)*w, clamp gradients are inclusive at the bounds (torch.clamp)
    float gb = 0.f;
    if (pt >= 1e-6f && pt <= 1.f - 1e-6f && sum >= 0.f && sum <= 1.f)
      gb = -cw * (2.f * t - 1.f) / ptc * (k2 * t + k3 * (1.f - t));
    gcle[i] = k0 * p_ * L * L - k1 * L + gb;
    gpse[i] = k0 * c * L * L - k1 * L + gb;
  }
}

// ----------------------------------------------------------------------------- reg-loss tail
// One block closes the whole train loss of models.py:549-574 from the seg-loss block sums: the fixed-order fp64
// fold of partial[nblk][6], dice + class-balanced BCE (metrics.py:10-37), both interval losses (models.py:512-521,
// band lookup of models.py:492-510 from the label), the total, and everything the backward needs (the four
// seg-loss coefficients for d loss, the d loss / d reg_outs rows).  Replaces ~90 O(B) torch launches.
struct RegTail {
  const float* partial; int nblk;
  const float* reg[2]; const long long* label[2]; const float* weight[2]; const float* bands[2]; int nband[2];
  int B; double N, smooth, beta, gamma;
  float* out;     // [5]  loss, loss_cle, loss_pse, mul, seg
  float* coef;    // [8]  seg-loss backward coefficients of d loss (g_mul = 2, g_seg = 1)
  float* greg;    // [2][B]
};

__global__ __launch_bounds__(256) void regloss_tail_kernel(RegTail a) {
  __shared__ double red[256];
  __shared__ double tot[8];
  const int tid = threadIdx.x;
  for (int k = 0; k < 6; ++k) {
    double s = 0.0;
    for (int r = tid; r < a.nblk; r += 256) s += (double)a.partial[(long)r * 6 + k];
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    if (tid == 0) tot[k] = red[0];
    __syncthreads();
  }
  for (int h = 0; h < 2; ++h) {
    double s = 0.0;
    for (int i = tid; i < a.B; i += 256) {
      const long long lab = a.label[h][i];
      double l = nan(""), g = nan("");            // a label outside the band table poisons the loss
      if (lab >= 0 && lab < a.nband[h]) {
        const double o = (double)a.reg[h][i], w = (double)a.weight[h][i];
        const double no = a.beta * pow(o, a.gamma);
        const double nl = a.beta * pow((double)a.bands[h][2 * lab], a.gamma);
        const double nh = a.beta * pow((double)a.bands[h][2 * lab + 1], a.gamma);
        const double hw = 0.5 * (nh - nl), d = no - 0.5 * (nh + nl);
        const double unh = d * d - hw * hw;
        l = unh > 0.0 ? 10.0 * unh * w : 0.0;
        g = unh > 0.0 ? 20.0 * w * d * a.beta * a.gamma * pow(o, a.gamma - 1.0) : 0.0;
      }
      a.greg[(long)h * a.B + i] = (float)g;
      s += l;
    }
    red[tid] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
      if (tid < w) red[tid] += red[tid + w];
      __syncthreads();
    }
    if (tid == 0) tot[6 + h] = red[0];
    __syncthreads();
  }
  if (tid == 0) {
    const double st = tot[0], A1 = tot[1], A0 = tot[2], I = tot[3], S1 = tot[4], S2 = tot[5];
    double alpha = 1.0 - st / (double)a.B;                      // metrics.py:18
    alpha = alpha < 0.3 ? 0.3 : (alpha > 0.7 ? 0.7 : alpha);
    const double sw = alpha * st + (1.0 - alpha) * (a.N - st);
    const double seg = (alpha * A1 + (1.0 - alpha) * A0) / sw;
    const double den = S1 + S2 + a.smooth;
    const double mul = (2.0 * I + a.smooth) / den;
    // float-rounded terms summed in float, as the reference's fp32 scalars are (models.py:572)
    const float lc = (float)tot[6], lp = (float)tot[7], fm = (float)mul, fs = (float)seg;
    a.out[0] = lc + lp + 2.0f * fm + fs;
    a.out[1] = lc; a.out[2] = lp; a.out[3] = fm; a.out[4] = fs;
    a.coef[0] = (float)(2.0 * 2.0 / den);
    a.coef[1] = (float)(2.0 * (2.0 * I + a.smooth) / (den * den));
    a.coef[2] = (float)(alpha / sw);
    a.coef[3] = (float)((1.0 - alpha) / sw);
    a.coef[4] = a.coef[5] = a.coef[6] = a.coef[7] = 0.f;
  }
}

inline NearGeom make_near(int Dl, int Hl, int Wl, int D, int H, int W) {
  NearGeom n;
  n.Dl = Dl; n.Hl = Hl; n.Wl = Wl;
  n.sz = (float)Dl / (float)D; n.sy = (float)Hl / (float)H; n.sx = (float)Wl / (float)W;
  return n;
}

inline int head_blocks(long long vps) {
  long long b = (vps + 2047) / 2048;
  return (int)(b > 512 ? 512 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int dram_head_nblk(long long voxels_per_sample) { return head_blocks(voxels_per_sample); }
extern "C" int dram_head_bwd_nparts(long long voxels_per_sample) { return head_blocks(voxels_per_sample); }

template <typename T>
static int head_fwd_impl(const T* x, const float* w, const float* bias, const float* lungs, int Dl, int Hl, int Wl,
                         float* dense, float* partial, int B, int D, int H, int W, int NO, int sigmoid,
                         dram_stream_t stream) {
  if (!x || !w || !bias || !dense || !partial || B < 1 || D < 1 || H < 1 || W < 1 || NO < 1 || NO > 16)
    return DRAM_ERR_BAD_ARG;
  if (lungs && (Dl < 1 || Hl < 1 || Wl < 1)) return DRAM_ERR_BAD_ARG;
  const long vps = (long)D * H * W;
  const int nblk = head_blocks(vps);
  const NearGeom ng = make_near(lungs ? Dl : 1, lungs ? Hl : 1, lungs ? Wl : 1, D, H, W);
  dim3 grid(nblk, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  DramProf prof(DRAM_FAM_HEAD_LOSS, 0, 0.0, (double)B * vps * (32.0 * sizeof(T) + 4.0 * (NO + (lungs ? 0.125 : 0.0))), s);
  if (NO <= 2)
    hipLaunchKernelGGL((head_fwd_kernel<2, T>), grid, block, 0, s, x, w, bias, lungs, ng, dense, partial, D, H, W, NO, sigmoid, nblk);
  else if (NO <= 9)
    hipLaunchKernelGGL((head_fwd_kernel<9, T>), grid, block, 0, s, x, w, bias, lungs, ng, dense, partial, D, H, W, NO, sigmoid, nblk);
  else
    hipLaunchKernelGGL((head_fwd_kernel<16, T>), grid, block, 0, s, x, w, bias, lungs, ng, dense, partial, D, H, W, NO, sigmoid, nblk);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_head_fwd(const float* x, const float* w, const float* bias, const float* lungs, int Dl, int Hl,
                             int Wl, float* dense, float* partial, int B, int D, int H, int W, int NO, int sigmoid,
                             dram_stream_t stream) {
  return head_fwd_impl<float>(x, w, bias, lungs, Dl, Hl, Wl, dense, partial, B, D, H, W, NO, sigmoid, stream);
}
extern "C" int dram_head_fwd_bf16(const void* x, const float* w, const float* bias, const float* lungs, int Dl, int Hl,
                                  int Wl, float* dense, float* partial, int B, int D, int H, int W, int NO, int sigmoid,
                                  dram_stream_t stream) {
  return head_fwd_impl<bf16_t>((const bf16_t*)x, w, bias, lungs, Dl, Hl, Wl, dense, partial, B, D, H, W, NO, sigmoid,
                               stream);
}

template <typename T>
static int head_bwd_impl(const T* x, const float* w, const float* dense, const float* gdense, const float* gpool,
                         const float* lungs, int Dl, int Hl, int Wl, T* dx, float* wpartial, int B, int D, int H, int W,
                         int NO, int sigmoid, dram_stream_t stream) {
  if (!x || !w || !gpool || !dx || !wpartial || B < 1 || D < 1 || H < 1 || W < 1 || NO < 1 || NO > 16)
    return DRAM_ERR_BAD_ARG;
  if (sigmoid && !dense) return DRAM_ERR_BAD_ARG;
  if (lungs && (Dl < 1 || Hl < 1 || Wl < 1)) return DRAM_ERR_BAD_ARG;
  const long vps = (long)D * H * W;
  const int nblk = head_blocks(vps);
  const NearGeom ng = make_near(lungs ? Dl : 1, lungs ? Hl : 1, lungs ? Wl : 1, D, H, W);
  dim3 grid(nblk, B), block(256);
  hipStream_t s = (hipStream_t)stream;
  DramProf prof(DRAM_FAM_HEAD_LOSS, 1, 0.0,
                (double)B * vps * (64.0 * sizeof(T) + 4.0 * ((dense ? NO : 0) + (gdense ? NO : 0) + (lungs ? 0.125 : 0.0))), s);
  if (NO <= 2)
    hipLaunchKernelGGL((head_bwd_kernel<2, T>), grid, block, 0, s, x, w, dense, gdense, gpool, lungs, ng, dx, wpartial, D, H, W, NO, sigmoid, nblk);
  else if (NO <= 9)
    hipLaunchKernelGGL((head_bwd_kernel<9, T>), grid, block, 0, s, x, w, dense, gdense, gpool, lungs, ng, dx, wpartial, D, H, W, NO, sigmoid, nblk);
  else
    hipLaunchKernelGGL((head_bwd_kernel<16, T>), grid, block, 0, s, x, w, dense, gdense, gpool, lungs, ng, dx, wpartial, D, H, W, NO, sigmoid, nblk);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_head_bwd(const float* x, const float* w, const float* dense, const float* gdense,
                             const float* gpool, const float* lungs, int Dl, int Hl, int Wl, float* dx,
                             float* wpartial, int B, int D, int H, int W, int NO, int sigmoid, dram_stream_t stream) {
  return head_bwd_impl<float>(x, w, dense, gdense, gpool, lungs, Dl, Hl, Wl, dx, wpartial, B, D, H, W, NO, sigmoid, stream);
}
extern "C" int dram_head_bwd_bf16(const void* x, const float* w, const float* dense, const float* gdense,
                                  const float* gpool, const float* lungs, int Dl, int Hl, int Wl, void* dx,
                                  float* wpartial, int B, int D, int H, int W, int NO, int sigmoid, dram_stream_t stream) {
  return head_bwd_impl<bf16_t>((const bf16_t*)x, w, dense, gdense, gpool, lungs, Dl, Hl, Wl, (bf16_t*)dx, wpartial, B, D, H,
                               W, NO, sigmoid, stream);
}

extern "C" int dram_segloss_nblk(long long voxels_total) {
  long long b = (voxels_total + 4095) / 4096;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" int dram_segloss_fwd(const float* cle, const float* pse, const float* lungs, const float* ems,
                                const float* binary, int Dl, int Hl, int Wl, float* partial, int B, int D, int H,
                                int W, float smoothness, dram_stream_t stream) {
  if (!cle || !pse || !lungs || !ems || !binary || !partial || B < 1 || D < 1 || H < 1 || W < 1 || Dl < 1 || Hl < 1 ||
      Wl < 1)
    return DRAM_ERR_BAD_ARG;
  const long total = (long)B * D * H * W;
  DramProf prof(DRAM_FAM_HEAD_LOSS, 2, 0.0, 4.0 * (double)total * 2.25, (hipStream_t)stream);   // cle, pse + 1/8 of 2 masks
  hipLaunchKernelGGL(segloss_fwd_kernel, dim3(dram_segloss_nblk(total)), dim3(256), 0, (hipStream_t)stream, cle, pse,
                     lungs, ems, binary, make_near(Dl, Hl, Wl, D, H, W), partial, B, D, H, W, smoothness);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_segloss_bwd(const float* cle, const float* pse, const float* lungs, const float* ems,
                                const float* binary, int Dl, int Hl, int Wl, const float* coef, float* gcle,
                                float* gpse, int B, int D, int H, int W, float smoothness, dram_stream_t stream) {
  if (!cle || !pse || !lungs || !ems || !binary || !coef || !gcle || !gpse || B < 1 || D < 1 || H < 1 || W < 1 ||
      Dl < 1 || Hl < 1 || Wl < 1)
    return DRAM_ERR_BAD_ARG;
  const long total = (long)B * D * H * W;
  DramProf prof(DRAM_FAM_HEAD_LOSS, 3, 0.0, 4.0 * (double)total * 4.25, (hipStream_t)stream);
  hipLaunchKernelGGL(segloss_bwd_kernel, dim3(dram_segloss_nblk(total)), dim3(256), 0, (hipStream_t)stream, cle, pse,
                     lungs, ems, binary, make_near(Dl, Hl, Wl, D, H, W), coef, gcle, gpse, B, D, H, W, smoothness);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_regloss_tail(const float* partial, int nblk, const float* reg_cle, const float* reg_pse,
                                 const long long* cle_labels, const long long* pse_labels, const float* cle_w,
                                 const float* pse_w, const float* cle_bands, int n_cle, const float* pse_bands,
                                 int n_pse, int B, double voxels_total, double smooth, double beta, double gamma,
                                 float* out, float* coef, float* greg, dram_stream_t stream) {
  if (!partial || nblk < 1 || !reg_cle || !reg_pse || !cle_labels || !pse_labels || !cle_w || !pse_w || !cle_bands ||
      !pse_bands || n_cle < 1 || n_pse < 1 || B < 1 || !out || !coef || !greg)
    return DRAM_ERR_BAD_ARG;
  RegTail a;
  a.partial = partial; a.nblk = nblk;
  a.reg[0] = reg_cle; a.reg[1] = reg_pse;
  a.label[0] = cle_labels; a.label[1] = pse_labels;
  a.weight[0] = cle_w; a.weight[1] = pse_w;
  a.bands[0] = cle_bands; a.bands[1] = pse_bands;
  a.nband[0] = n_cle; a.nband[1] = n_pse;
  a.B = B; a.N = voxels_total; a.smooth = smooth; a.beta = beta; a.gamma = gamma;
  a.out = out; a.coef = coef; a.greg = greg;
  hipLaunchKernelGGL(regloss_tail_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, a);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
