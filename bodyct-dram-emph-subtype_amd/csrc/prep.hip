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


// ---------------------------------------------------------------------------------------------------
// Post-processing of the predict path (reference processor.py:111-129, :143): the dRAM volume [D,H,W] is
// resized (trilinear, align_corners=True) to the lung-crop size and pasted into a zero volume of the
// original scan grid; optionally also written as uint8 through windowing(0..1 -> 0..255) + truncation.
// One pass over the ORIGINAL grid (gather form).
__global__ void resample_paste_kernel(const float* __restrict__ src, float* __restrict__ outf,
                                      uint8_t* __restrict__ outb, int D, int H, int W, int rd, int rh, int rw, int oz,
                                      int oy, int ox, int Do, int Ho, int Wo, float sz, float sy, float sx) {
  const long total = (long)Do * Ho * Wo;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int x = (int)(r % Wo) - ox; r /= Wo;
    const int y = (int)(r % Ho) - oy;
    const int z = (int)(r / Ho) - oz;
    float v = 0.f;
    if (z >= 0 && z < rd && y >= 0 && y < rh && x >= 0 && x < rw) {
      const float fz = sz * (float)z, fy = sy * (float)y, fx = sx * (float)x;   // ATen area_pixel_compute_source_index
      int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
      if (z0 > D - 1) z0 = D - 1;
      if (y0 > H - 1) y0 = H - 1;
      if (x0 > W - 1) x0 = W - 1;
      const int z1 = z0 + (z0 < D - 1 ? 1 : 0), y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
      const float wz1 = fz - (float)z0, wy1 = fy - (float)y0, wx1 = fx - (float)x0;
      const float wz0 = 1.f - wz1, wy0 = 1.f - wy1, wx0 = 1.f - wx1;
#define RP_AT(zz, yy, xx) src[((long)(zz) * H + (yy)) * W + (xx)]
      v = wz0 * wy0 * wx0 * RP_AT(z0, y0, x0) + wz0 * wy0 * wx1 * RP_AT(z0, y0, x1) +
          wz0 * wy1 * wx0 * RP_AT(z0, y1, x0) + wz0 * wy1 * wx1 * RP_AT(z0, y1, x1) +
          wz1 * wy0 * wx0 * RP_AT(z1, y0, x0) + wz1 * wy0 * wx1 * RP_AT(z1, y0, x1) +
          wz1 * wy1 * wx0 * RP_AT(z1, y1, x0) + wz1 * wy1 * wx1 * RP_AT(z1, y1, x1);
#undef RP_AT
    }
    if (outf) outf[i] = v;
    if (outb) {   // utils.windowing(full, from_span=(0, 1)) in float64, then .astype(np.uint8) (truncation)
      double w = (double)v;
      w = w < 0.0 ? 0.0 : (w > 1.0 ? 1.0 : w);
      outb[i] = (uint8_t)(w * 255.0);
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// Train-time augmentations of the reference data module (models.py:66-74) with GIVEN parameters:
//   GaussianAddictive (intensity_transforms.py:145-177): rescale to 0..1 by the volume min / range, add
//     sigma * noise, clip, rescale back;  BoxMaskOut (:180-237): boxes set to 0;  Flip
//     (spatial_transforms.py:100-131): torch.flip over a set of axes;  CropAndResize (:133-197 + functional.py
//     roi_align): affine_grid (align_corners=False base grid) + grid_sample (image: trilinear, zero padding,
//     align_corners=True; mask: nearest, align_corners=False) of the normalised bounding box.
// All four are fused into ONE gather pass per output voxel: the (up to) 8 sampled source voxels are un-flipped,
// box-tested and noised on the fly -- the intermediate volumes are never written.
struct AugParams {
  int flags;           // bit 0 noise, 1 boxes, 2 flip, 3 crop-resize
  int nbox;
  int box[10][6];      // z0,z1,y0,y1,x0,x1 (half-open) in the pre-flip grid
  int flip;            // bit 0: flip z, 1: y, 2: x
  float sigma;
  float blo[3], bhi[3];  // normalised bounding box (lo/size, hi/size) for z, y, x
};

__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, float* __restrict__ partial, long n) {
  __shared__ float sm[2][4];
  float lo = INFINITY, hi = -INFINITY;
  for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += (long)gridDim.x * 256L) {
    const float v = x[i];
    lo = fminf(lo, v);
    hi = fmaxf(hi, v);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) { sm[0][threadIdx.x >> 6] = lo; sm[1][threadIdx.x >> 6] = hi; }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[blockIdx.x * 2] = fminf(fminf(sm[0][0], sm[0][1]), fminf(sm[0][2], sm[0][3]));
    partial[blockIdx.x * 2 + 1] = fmaxf(fmaxf(sm[1][0], sm[1][1]), fmaxf(sm[1][2], sm[1][3]));
  }
}

// value of the volume AFTER noise, boxes and flip at (z, y, x) of the flipped grid
__device__ __forceinline__ float aug_source(const float* __restrict__ x, const float* __restrict__ noise,
                                            const AugParams& p, const float dmin, const float drange, const float inv,
                                            int z, int y, int xx, const int D, const int H, const int W) {
  if (p.flags & 4) {
    if (p.flip & 1) z = D - 1 - z;
    if (p.flip & 2) y = H - 1 - y;
    if (p.flip & 4) xx = W - 1 - xx;
  }
  if (p.flags & 2)
    for (int b = 0; b < p.nbox; ++b)
      if (z >= p.box[b][0] && z < p.box[b][1] && y >= p.box[b][2] && y < p.box[b][3] && xx >= p.box[b][4] &&
          xx < p.box[b][5])
        return 0.f;
  const long o = ((long)z * H + y) * W + xx;
  float v = x[o];
  if (p.flags & 1) {
    float r = (v - dmin) / inv + p.sigma * noise[o];     // inv = float(d_range + 1e-7)
    r = r < 0.f ? 0.f : (r > 1.f ? 1.f : r);
    v = r * drange + dmin;
  }
  return v;
}

__global__ void augment_image_kernel(const float* __restrict__ x, const float* __restrict__ noise,
                                     const float* __restrict__ mm, float* __restrict__ out, const AugParams p,
                                     const int D, const int H, const int W) {
  const long total = (long)D * H * W;
  const float dmin = mm[0], drange = mm[1] - mm[0];
  const float inv = drange + 1e-7f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    const int xo = (int)(r % W); r /= W;
    const int yo = (int)(r % H);
    const int zo = (int)(r / H);
    if (!(p.flags & 8)) {
      out[i] = aug_source(x, noise, p, dmin, drange, inv, zo, yo, xo, D, H, W);
      continue;
    }
    // affine_grid base coordinate (align_corners=False): u = (2k + 1)/S - 1; c = (hi - lo) * u + (lo + hi - 1);
    // grid_sample align_corners=True: pixel = (c + 1)/2 * (S - 1)
    float pix[3];
    const int kk[3] = {zo, yo, xo}, S[3] = {D, H, W};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const float u = (2.f * (float)kk[a] + 1.f) / (float)S[a] - 1.f;
      const float c = (p.bhi[a] - p.blo[a]) * u + (p.blo[a] + p.bhi[a] - 1.f);
      pix[a] = (c + 1.f) * 0.5f * (float)(S[a] - 1);
    }
    const float fz = floorf(pix[0]), fy = floorf(pix[1]), fx = floorf(pix[2]);
    const int z0 = (int)fz, y0 = (int)fy, x0 = (int)fx;
    const float wz1 = pix[0] - fz, wy1 = pix[1] - fy, wx1 = pix[2] - fx;
    float acc = 0.f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          const int z = z0 + dz, y = y0 + dy, xx = x0 + dx;
          if (z < 0 || z >= D || y < 0 || y >= H || xx < 0 || xx >= W) continue;     // padding_mode='zeros'
          const float w = (dz ? wz1 : 1.f - wz1) * (dy ? wy1 : 1.f - wy1) * (dx ? wx1 : 1.f - wx1);
          acc += w * aug_source(x, noise, p, dmin, drange, inv, z, y, xx, D, H, W);
        }
    out[i] = acc;
  }
}

// masks: Flip + CropAndResize(nearest, align_corners=False, zero padding)
__global__ void augment_mask_kernel(const float* __restrict__ m, float* __restrict__ out, const AugParams p,
                                    const int D, const int H, const int W) {
  const long total = (long)D * H * W;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long r = i;
    int xo = (int)(r % W); r /= W;
    int yo = (int)(r % H);
    int zo = (int)(r / H);
    bool inside = true;
    if (p.flags & 8) {
      int q[3];
      const int kk[3] = {zo, yo, xo}, S[3] = {D, H, W};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float u = (2.f * (float)kk[a] + 1.f) / (float)S[a] - 1.f;
        const float c = (p.bhi[a] - p.blo[a]) * u + (p.blo[a] + p.bhi[a] - 1.f);
        const float pix = ((c + 1.f) * (float)S[a] - 1.f) * 0.5f;        // align_corners=False
        q[a] = (int)nearbyintf(pix);                                      // round half to even, as ATen
        inside = inside && q[a] >= 0 && q[a] < S[a];
      }
      zo = q[0]; yo = q[1]; xo = q[2];
    }
    float v = 0.f;
    if (inside) {
      if (p.flags & 4) {
        if (p.flip & 1) zo = D - 1 - zo;
        if (p.flip & 2) yo = H - 1 - yo;
        if (p.flip & 4) xo = W - 1 - xo;
      }
      v = m[((long)zo * H + yo) * W + xo];
    }
    out[i] = v;
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

extern "C" int dram_resample_paste(const float* src, float* out_f32, uint8_t* out_u8, int D, int H, int W, int rd, int rh,
                                   int rw, int oz, int oy, int ox, int Do, int Ho, int Wo, dram_stream_t stream) {
  if (!src || (!out_f32 && !out_u8) || D < 1 || H < 1 || W < 1 || rd < 1 || rh < 1 || rw < 1 || oz < 0 || oy < 0 ||
      ox < 0 || oz + rd > Do || oy + rh > Ho || ox + rw > Wo)
    return DRAM_ERR_BAD_ARG;
  auto sc = [](int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; };
  const long total = (long)Do * Ho * Wo;
  DramProf prof(DRAM_FAM_PREP, 3, 0.0, 4.0 * (double)D * H * W + (double)total * ((out_f32 ? 4 : 0) + (out_u8 ? 1 : 0)),
                (hipStream_t)stream);
  hipLaunchKernelGGL(resample_paste_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, src, out_f32, out_u8,
                     D, H, W, rd, rh, rw, oz, oy, ox, Do, Ho, Wo, sc(D, rd), sc(H, rh), sc(W, rw));
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_minmax_nblk(long long n) { return dram_window_stats_nblk(n); }

extern "C" int dram_minmax(const float* x, float* partial, long long n, dram_stream_t stream) {
  if (!x || !partial || n < 1) return DRAM_ERR_BAD_ARG;
  DramProf prof(DRAM_FAM_PREP, 4, 0.0, 4.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(minmax_kernel, dim3(dram_minmax_nblk(n)), dim3(256), 0, (hipStream_t)stream, x, partial, (long)n);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

static int fill_aug(AugParams& p, const DramAugment* a) {
  if (!a || a->n_boxes < 0 || a->n_boxes > 10) return DRAM_ERR_BAD_ARG;
  p.flags = a->flags & 15;
  p.nbox = a->n_boxes;
  for (int b = 0; b < a->n_boxes; ++b)
    for (int k = 0; k < 6; ++k) p.box[b][k] = a->boxes[b][k];
  p.flip = a->flip_axes & 7;
  p.sigma = a->sigma;
  for (int k = 0; k < 3; ++k) { p.blo[k] = a->box_lo[k]; p.bhi[k] = a->box_hi[k]; }
  return DRAM_OK;
}

extern "C" int dram_augment_image(const float* x, const float* noise, const float* minmax, float* out, int D, int H,
                                  int W, const DramAugment* aug, dram_stream_t stream) {
  if (!x || !out || x == out || D < 1 || H < 1 || W < 1) return DRAM_ERR_BAD_ARG;
  AugParams p{};
  const int rc = fill_aug(p, aug);
  if (rc != DRAM_OK) return rc;
  if ((p.flags & 1) && (!noise || !minmax)) return DRAM_ERR_BAD_ARG;
  static const float zero2[2] = {0.f, 0.f};
  (void)zero2;
  const long total = (long)D * H * W;
  DramProf prof(DRAM_FAM_PREP, 5, 0.0, 4.0 * (double)total * ((p.flags & 1) ? 3.0 : 2.0), (hipStream_t)stream);
  hipLaunchKernelGGL(augment_image_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, x,
                     (p.flags & 1) ? noise : x, (p.flags & 1) ? minmax : x, out, p, D, H, W);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_augment_mask(const float* mask, float* out, int D, int H, int W, const DramAugment* aug,
                                 dram_stream_t stream) {
  if (!mask || !out || mask == out || D < 1 || H < 1 || W < 1) return DRAM_ERR_BAD_ARG;
  AugParams p{};
  const int rc = fill_aug(p, aug);
  if (rc != DRAM_OK) return rc;
  const long total = (long)D * H * W;
  DramProf prof(DRAM_FAM_PREP, 6, 0.0, 8.0 * (double)total, (hipStream_t)stream);
  hipLaunchKernelGGL(augment_mask_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, mask, out, p, D, H, W);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
