// prep.hip -- GPU-side deterministic input transforms of the reference's data module:
// IntensityWindow(-1150..-300 -> 0..1), Standardize (volume z-score, unbiased std) and
// Interpolate(target_size, align_corners=True, only_in_plane=True): bilinear in-plane resize +
// depth index selection torch.linspace(0, D-1, newD).long()  (reference models.py:59-63,
// functional.py:13-26, intensity_transforms.py:108-111, spatial_transforms.py:55-75;
// masks: nearest in-plane + the same depth indices, spatial_transforms.py:77-98).
// One reduction pass over the raw scan + one fused window/standardize/resize pass per output
// voxel (the z-score is linear, so it commutes with the bilinear weights).  HBM-bound.
#include "common.h"

namespace {

__device__ __forceinline__ float window01(float v, float lo, float hi) {
  v = fminf(fmaxf(v, lo), hi);
  return (v - lo) / (hi - lo);
}

// partial[blk][0] = sum w, partial[blk][1] = sum w*w  over one volume (w = windowed value)
__global__ __launch_bounds__(256) void window_stats_kernel(const float* __restrict__ scan, float* __restrict__ partial,
                                                           long n, float lo, float hi) {
  __shared__ float sm[2][4];
  float s0 = 0.f, s1 = 0.f;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const float w = window01(scan[i], lo, hi);
    s0 += w;
    s1 += w * w;
  }
  s0 = wave_sum(s0);
  s1 = wave_sum(s1);
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = s0; sm[1][threadIdx.x >> 6] = s1; }
  __syncthreads();
  if (threadIdx.x < 2)
    partial[blockIdx.x * 2 + threadIdx.x] = sm[threadIdx.x][0] + sm[threadIdx.x][1] + sm[threadIdx.x][2] + sm[threadIdx.x][3];
}

// out[z][y][x] = (bilinear_{align_corners}(window(scan[zidx[z]]))(y, x) - mean) * inv_std
__global__ void prep_image_kernel(const float* __restrict__ scan, const int* __restrict__ zidx,
                                  const float* __restrict__ mean_invstd, float* __restrict__ out, int H, int W,
                                  int Do, int Ho, int Wo, float sy, float sx, float lo, float hi) {
  const long total = (long)Do * Ho * Wo;
  const float mean = mean_invstd[0], inv = mean_invstd[1];
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int zo = (int)(r / Ho);
    const float fy = sy * (float)yo, fx = sx * (float)xo;
    int y0 = (int)fy, x0 = (int)fx;
    if (y0 > H - 1) y0 = H - 1;
    if (x0 > W - 1) x0 = W - 1;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float wy1 = fy - (float)y0, wx1 = fx - (float)x0;
    const float wy0 = 1.f - wy1, wx0 = 1.f - wx1;
    const float* p = scan + (long)zidx[zo] * H * W;
    const float v = wy0 * (wx0 * window01(p[(long)y0 * W + x0], lo, hi) + wx1 * window01(p[(long)y0 * W + x1], lo, hi)) +
                    wy1 * (wx0 * window01(p[(long)y1 * W + x0], lo, hi) + wx1 * window01(p[(long)y1 * W + x1], lo, hi));
    out[i] = (v - mean) * inv;
  }
}

// nearest in-plane (F.interpolate 'nearest': src = min(floor(dst * in/out), in-1)) + depth select
__global__ void prep_mask_kernel(const float* __restrict__ mask, const int* __restrict__ zidx, float* __restrict__ out,
                                 int H, int W, int Do, int Ho, int Wo, float sy, float sx) {
  const long total = (long)Do * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int zo = (int)(r / Ho);
    int ys = (int)floorf((float)yo * sy), xs = (int)floorf((float)xo * sx);
    if (ys > H - 1) ys = H - 1;
    if (xs > W - 1) xs = W - 1;
    out[i] = mask[((long)zidx[zo] * H + ys) * W + xs];
  }
}

inline int grid_for(long n) {
  long b = (n + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" int dram_window_stats_nblk(long long n) {
  long long b = (n + 4095) / 4096;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" int dram_window_stats(const float* scan, float* partial, long long n, float lo, float hi,
                                 dram_stream_t stream) {
  if (!scan || !partial || n < 2 || !(hi > lo)) return DRAM_ERR_BAD_ARG;
  DramProf prof(DRAM_FAM_PREP, 0, 0.0, 4.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(window_stats_kernel, dim3(dram_window_stats_nblk(n)), dim3(256), 0, (hipStream_t)stream, scan,
                     partial, (long)n, lo, hi);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_prep_image(const float* scan, const int* zidx, const float* mean_invstd, float* out, int D, int H,
                               int W, int Do, int Ho, int Wo, float lo, float hi, dram_stream_t stream) {
  if (!scan || !zidx || !mean_invstd || !out || D < 1 || H < 1 || W < 1 || Do < 1 || Ho < 1 || Wo < 1 || !(hi > lo))
    return DRAM_ERR_BAD_ARG;
  const float sy = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
  const float sx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
  DramProf prof(DRAM_FAM_PREP, 1, 0.0, 4.0 * ((double)Do * H * W + (double)Do * Ho * Wo), (hipStream_t)stream);
  hipLaunchKernelGGL(prep_image_kernel, dim3(grid_for((long)Do * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, scan,
                     zidx, mean_invstd, out, H, W, Do, Ho, Wo, sy, sx, lo, hi);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_prep_mask(const float* mask, const int* zidx, float* out, int D, int H, int W, int Do, int Ho,
                              int Wo, dram_stream_t stream) {
  if (!mask || !zidx || !out || D < 1 || H < 1 || W < 1 || Do < 1 || Ho < 1 || Wo < 1) return DRAM_ERR_BAD_ARG;
  DramProf prof(DRAM_FAM_PREP, 2, 0.0, 4.0 * 2.0 * (double)Do * Ho * Wo, (hipStream_t)stream);
  hipLaunchKernelGGL(prep_mask_kernel, dim3(grid_for((long)Do * Ho * Wo)), dim3(256), 0, (hipStream_t)stream, mask,
                     zidx, out, H, W, Do, Ho, Wo, (float)H / (float)Ho, (float)W / (float)Wo);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
