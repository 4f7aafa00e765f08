// upmix.hip -- the first decoder convolution WITHOUT the up-sampled tensor: channel mixing at LOW resolution.
//
// Reference call site: med3d.py:83-87 (UpsampleConvBlock5d.forward: upsample x2 trilinear align_corners, crop_concat_5d,
// conv_blocks[0] = Conv3d 3x3x3 pad 1) for us1, whose up-sampled operand carries 512 (ResNet-18/34) or 2 048
// (ResNet-50) of the 576 / 2 304 input channels.  With c = concat(up(a), s):
//
//     conv(c)[v] = sum_t W_up[t] . up(a)[v + t - 1]  +  conv_s(s)[v]            (t = 27 taps, zero outside the volume)
//
// up() acts per channel and W_up[t] mixes channels per voxel, so they commute:  W_up[t] . up(a) = up(W_up[t] . a).
// The channel mixing therefore runs at the LOW resolution -- one plain GEMM a[n][Cu] -> b[n][27 Co] over an eighth of
// the voxels (8x fewer FLOPs than the convolution of the up-sampled tensor, which is never built) -- and what is left
// at the high resolution is a gather of 27 taps x 8 trilinear corners of a Co-channel tensor.  That gather is
// separable per axis:   out[vz][vy][vx] = sum_{kz,cz} wz ( sum_{ky,cy} wy ( sum_{kx,cx} wx b[iz][iy][ix][kz][ky][kx] ) )
// so it runs as three 1-D passes of ONE generic kernel (6 terms per output each; 28.5 FMAs per output voxel and
// channel instead of 216):
//     In[outer][n < Ni][m][k < 3][C]  ->  Out[outer][v < 2 Ni][m][C],
//     Out[o][v][m][c] = sum_k [0 <= u = v + k - 1 < 2 Ni] ( w0(u) In[o][i0(u)][m][k][c] + w1(u) In[o][i1(u)][m][k][c] )
//   pass X: outer = (B, nz, ny), m = (kz, ky);   pass Y: outer = (B, nz), m = (vx, kz);   pass Z: outer = B, m = (vy, vx)
// (the tap index of b is t = (kz * 3 + ky) * 3 + kx: each pass contracts the innermost remaining tap).  The last pass
// adds the skip convolution's result (bias included), stores in the storage type and takes the BatchNorm partial sums
// of the stored values, like a convolution epilogue.  Backward = the exact transposes in reverse order (a gather per
// low-resolution element: no atomics), followed by the 1x1x1 data / weight gradient GEMMs at the low resolution.
// Intermediates between the passes are fp32 whatever the storage type.  Same interpolation arithmetic as
// upcat_fwd_kernel (lin_src: PyTorch's align_corners source index).
#include "common.h"

namespace {

__device__ __forceinline__ float4 f4_fma(const float w, const float4 a, const float4 acc) {
  return make_float4(fmaf(w, a.x, acc.x), fmaf(w, a.y, acc.y), fmaf(w, a.z, acc.z), fmaf(w, a.w, acc.w));
}

// one output element (4 channels) of the generic pass
template <typename TIN>
__device__ __forceinline__ float4 axis_gather(const TIN* __restrict__ in, const long o, const int v, const int mi,
                                              const int c4, const int Ni, const int m, const int C, const float scale) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  // all six loads first, no branch in between (an out-of-range tap reads a clamped address with zero weights: a
  // per-lane `continue` around each pair kept two loads in flight)
  float4 x0[3], x1[3];
  float w0[3], w1[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int u = v + k - 1;
    const bool ok = u >= 0 && u < 2 * Ni;
    const int uc = u < 0 ? 0 : (u >= 2 * Ni ? 2 * Ni - 1 : u);
    int i0, i1;
    lin_src(uc, scale, Ni, i0, i1, w0[k], w1[k]);
    if (!ok) { w0[k] = 0.f; w1[k] = 0.f; }
    const long base = ((long)mi * 3 + k) * C + 4 * c4;
    const long row = (long)m * 3 * C;
    x0[k] = ld4<TIN>(in, (o * Ni + i0) * row + base);
    x1[k] = ld4<TIN>(in, (o * Ni + i1) * row + base);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    acc = f4_fma(w0[k], x0[k], acc);
    acc = f4_fma(w1[k], x1[k], acc);
  }
  return acc;
}

template <typename TIN>
__global__ __launch_bounds__(256) void upmix_axis_fwd_kernel(const TIN* __restrict__ in, float* __restrict__ out,
                                                             const long total, const int Ni, const int m, const int C,
                                                             const float scale) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= total) return;
  const int C4 = C >> 2;
  const int c4 = (int)(i % C4);
  long r = i / C4;
  const int mi = (int)(r % m); r /= m;
  const int v = (int)(r % (2 * Ni));
  const long o = r / (2 * Ni);
  st4<float>(out, 4 * i, axis_gather<TIN>(in, o, v, mi, c4, Ni, m, C, scale));
}

// last pass: + base (the skip convolution's output, may alias out), stored as T, BatchNorm partial sums of the STORED
// values per block of `vpb` voxels: stats[(blk * 2 + {0: sum, 1: sum of squares}) * C + c]
template <typename T>
__global__ __launch_bounds__(256) void upmix_axis_fwd_final_kernel(const float* __restrict__ in, const T* base,
                                                                   T* out, float* __restrict__ stats, const long nvox,
                                                                   const int Ni, const int m, const int C,
                                                                   const float scale, const int vpb) {
  __shared__ float red[2][256][4];
  const int C4 = C >> 2;
  const int tpv = C4;                       // threads per voxel
  const int vpi = 256 / tpv;                // voxels per iteration
  const int c4 = threadIdx.x % tpv, vl = threadIdx.x / tpv;
  const long v0 = (long)blockIdx.x * vpb;
  const long v1 = v0 + vpb < nvox ? v0 + vpb : nvox;
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = s1;
  for (long vox = v0 + vl; vox < v1; vox += vpi) {
    const int mi = (int)(vox % m);
    const long r = vox / m;
    const int v = (int)(r % (2 * Ni));
    const long o = r / (2 * Ni);
    float4 a = axis_gather<float>(in, o, v, mi, c4, Ni, m, C, scale);
    const long off = vox * C + 4 * c4;
    if (base) {
      const float4 bv = ld4<T>(base, off);
      a = make_float4(a.x + bv.x, a.y + bv.y, a.z + bv.z, a.w + bv.w);
    }
    st4<T>(out, off, a);
    if (sizeof(T) == 2) {                   // statistics of the rounded values (what the consumer reads)
      a = make_float4(bf16_to_f32(f32_to_bf16(a.x)), bf16_to_f32(f32_to_bf16(a.y)), bf16_to_f32(f32_to_bf16(a.z)),
                      bf16_to_f32(f32_to_bf16(a.w)));
    }
    s1 = make_float4(s1.x + a.x, s1.y + a.y, s1.z + a.z, s1.w + a.w);
    s2 = make_float4(fmaf(a.x, a.x, s2.x), fmaf(a.y, a.y, s2.y), fmaf(a.z, a.z, s2.z), fmaf(a.w, a.w, s2.w));
  }
  if (!stats) return;
  red[0][threadIdx.x][0] = s1.x; red[0][threadIdx.x][1] = s1.y; red[0][threadIdx.x][2] = s1.z; red[0][threadIdx.x][3] = s1.w;
  red[1][threadIdx.x][0] = s2.x; red[1][threadIdx.x][1] = s2.y; red[1][threadIdx.x][2] = s2.z; red[1][threadIdx.x][3] = s2.w;
  __syncthreads();
  if ((int)threadIdx.x < 2 * C) {           // C <= 128
    const int which = threadIdx.x / C, c = threadIdx.x - which * C;
    float s = 0.f;
    for (int j = 0; j < vpi; ++j) s += red[which][j * tpv + (c >> 2)][c & 3];     // fixed order: deterministic
    stats[((long)blockIdx.x * 2 + which) * C + c] = s;
  }
}

// transposed pass:  H[o][n][m][k][c] = sum_u [i0(u) == n] w0(u) G[o][u - k + 1][m][c] + [i1(u) == n] w1(u) G[...]
// over the u in [0, 2 Ni) with 0 <= u - k + 1 < 2 Ni.  Candidates: scale * u in (n - 1, n + 1); the window below is one
// wider on both sides and every candidate is tested with the forward's own lin_src, so the two operators are exact
// transposes whatever the rounding of scale * u.
template <typename TG, typename TH>
__global__ __launch_bounds__(256) void upmix_axis_bwd_kernel(const TG* __restrict__ g, TH* __restrict__ h,
                                                             const long total, const int Ni, const int m, const int C,
                                                             const float scale, const float inv_scale) {
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= total) return;
  const int C4 = C >> 2;
  const int c4 = (int)(i % C4);
  long r = i / C4;
  const int k = (int)(r % 3); r /= 3;
  const int mi = (int)(r % m); r /= m;
  const int n = (int)(r % Ni);
  const long o = r / Ni;
  int ulo = 0, uhi = 2 * Ni - 1;
  if (scale > 0.f) {
    ulo = (int)((float)(n - 1) * inv_scale) - 1;
    uhi = (int)((float)(n + 1) * inv_scale) + 2;
    if (ulo < 0) ulo = 0;
    if (uhi > 2 * Ni - 1) uhi = 2 * Ni - 1;
  }
  // (a fixed, unrolled window of 8 candidates with zero weights instead of this loop -- every load issued up front --
  // measured SLOWER: 130 -> 145, 79 -> 98, 55 -> 68 us; about half the candidates are real, so it doubles the loads)
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int u = ulo; u <= uhi; ++u) {
    const int v = u - k + 1;
    if (v < 0 || v >= 2 * Ni) continue;
    int i0, i1;
    float w0, w1;
    lin_src(u, scale, Ni, i0, i1, w0, w1);
    const float w = (i0 == n ? w0 : 0.f) + (i1 == n ? w1 : 0.f);
    if (i0 != n && i1 != n) continue;
    acc = f4_fma(w, ld4<TG>(g, ((o * 2 * Ni + v) * m + mi) * (long)C + 4 * c4), acc);
  }
  st4<TH>(h, 4 * i, acc);
}

// w [Co][Cu + Cs][27]  ->  wlo [27 Co][Cu] (row t * Co + co: the low-resolution GEMM's weight, K = Cu contiguous),
//                          ws  [Co][Cs][27] (the skip channels' 3x3x3 weight)
__global__ __launch_bounds__(256) void upmix_split_weight_kernel(const float* __restrict__ w, float* __restrict__ wlo,
                                                                 float* __restrict__ ws, const int Co, const int Cu,
                                                                 const int Cs) {
  const long total = (long)Co * (Cu + Cs) * 27;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= total) return;
  const int t = (int)(i % 27);
  const long r = i / 27;
  const int ci = (int)(r % (Cu + Cs)), co = (int)(r / (Cu + Cs));
  const float v = w[i];
  if (ci < Cu) wlo[((long)t * Co + co) * Cu + ci] = v;
  else ws[((long)co * Cs + (ci - Cu)) * 27 + t] = v;
}

__global__ __launch_bounds__(256) void upmix_merge_wgrad_kernel(const float* __restrict__ dwlo,
                                                                const float* __restrict__ dws, float* __restrict__ dw,
                                                                const int Co, const int Cu, const int Cs) {
  const long total = (long)Co * (Cu + Cs) * 27;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= total) return;
  const int t = (int)(i % 27);
  const long r = i / 27;
  const int ci = (int)(r % (Cu + Cs)), co = (int)(r / (Cu + Cs));
  dw[i] = ci < Cu ? dwlo[((long)t * Co + co) * Cu + ci] : dws[((long)co * Cs + (ci - Cu)) * 27 + t];
}

inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

bool axis_ok(long long outer, int Ni, int m, int C) {
  if (outer < 1 || Ni < 1 || m < 1 || C < 4 || (C & 3)) return false;
  return outer * 2 * Ni * m * 3 * C < (1LL << 40);
}

template <typename TIN>
int axis_fwd(const TIN* in, float* out, long long outer, int Ni, int m, int C, hipStream_t s) {
  const long total = (long)outer * 2 * Ni * m * (C >> 2);
  DramProf prof(DRAM_FAM_POOL_UP, 10, 0.0, (double)total * 4.0 * (1.5 * sizeof(TIN) + 4.0), s);
  hipLaunchKernelGGL((upmix_axis_fwd_kernel<TIN>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, in, out, total,
                     Ni, m, C, ac_scale(Ni, 2 * Ni));
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

constexpr int UPMIX_VPB = 512;      // voxels per block (one BatchNorm partial row) of the last pass

template <typename T>
int axis_fwd_final(const float* in, const T* base, T* out, float* stats, long long outer, int Ni, int m, int C,
                   hipStream_t s) {
  if (C > 128 || 256 % (C >> 2)) return DRAM_ERR_UNSUPPORTED;
  const long nvox = (long)outer * 2 * Ni * m;
  DramProf prof(DRAM_FAM_POOL_UP, 11, 0.0, (double)nvox * C * (6.0 + sizeof(T) * (base ? 2.0 : 1.0)), s);
  hipLaunchKernelGGL((upmix_axis_fwd_final_kernel<T>), dim3((unsigned)((nvox + UPMIX_VPB - 1) / UPMIX_VPB)), dim3(256), 0,
                     s, in, base, out, stats, nvox, Ni, m, C, ac_scale(Ni, 2 * Ni), UPMIX_VPB);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

template <typename TG, typename TH>
int axis_bwd(const TG* g, TH* h, long long outer, int Ni, int m, int C, hipStream_t s) {
  const long total = (long)outer * Ni * m * 3 * (C >> 2);
  const float sc = ac_scale(Ni, 2 * Ni);
  DramProf prof(DRAM_FAM_POOL_UP, 12, 0.0, (double)total * 4.0 * (sizeof(TH) + sizeof(TG) * 2.0 / 3.0), s);
  hipLaunchKernelGGL((upmix_axis_bwd_kernel<TG, TH>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, g, h, total,
                     Ni, m, C, sc, sc > 0.f ? 1.f / sc : 0.f);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

}  // namespace

extern "C" int dram_upmix_stat_rows(long long nvox) { return (int)((nvox + UPMIX_VPB - 1) / UPMIX_VPB); }

extern "C" int dram_upmix_axis_fwd(const void* in, int in_bf16, float* out, long long outer, int Ni, int m, int C,
                                   dram_stream_t stream) {
  if (!in || !out) return DRAM_ERR_BAD_ARG;
  if (!axis_ok(outer, Ni, m, C)) return DRAM_ERR_UNSUPPORTED;
  return in_bf16 ? axis_fwd<bf16_t>((const bf16_t*)in, out, outer, Ni, m, C, (hipStream_t)stream)
                 : axis_fwd<float>((const float*)in, out, outer, Ni, m, C, (hipStream_t)stream);
}

extern "C" int dram_upmix_axis_fwd_final(const float* in, const void* base, void* out, int out_bf16, float* stats,
                                         long long outer, int Ni, int m, int C, dram_stream_t stream) {
  if (!in || !out) return DRAM_ERR_BAD_ARG;
  if (!axis_ok(outer, Ni, m, C)) return DRAM_ERR_UNSUPPORTED;
  return out_bf16 ? axis_fwd_final<bf16_t>(in, (const bf16_t*)base, (bf16_t*)out, stats, outer, Ni, m, C, (hipStream_t)stream)
                  : axis_fwd_final<float>(in, (const float*)base, (float*)out, stats, outer, Ni, m, C, (hipStream_t)stream);
}

extern "C" int dram_upmix_axis_bwd(const void* g, int g_bf16, void* h, int h_bf16, long long outer, int Ni, int m, int C,
                                   dram_stream_t stream) {
  if (!g || !h) return DRAM_ERR_BAD_ARG;
  if (!axis_ok(outer, Ni, m, C)) return DRAM_ERR_UNSUPPORTED;
  hipStream_t s = (hipStream_t)stream;
  if (g_bf16 && !h_bf16) return axis_bwd<bf16_t, float>((const bf16_t*)g, (float*)h, outer, Ni, m, C, s);
  if (!g_bf16 && h_bf16) return axis_bwd<float, bf16_t>((const float*)g, (bf16_t*)h, outer, Ni, m, C, s);
  if (g_bf16 && h_bf16) return axis_bwd<bf16_t, bf16_t>((const bf16_t*)g, (bf16_t*)h, outer, Ni, m, C, s);
  return axis_bwd<float, float>((const float*)g, (float*)h, outer, Ni, m, C, s);
}

extern "C" int dram_upmix_split_weight(const float* w, float* wlo, float* ws, int Co, int Cu, int Cs,
                                       dram_stream_t stream) {
  if (!w || !wlo || !ws || Co < 1 || Cu < 1 || Cs < 1) return DRAM_ERR_BAD_ARG;
  const long total = (long)Co * (Cu + Cs) * 27;
  DramProf prof(DRAM_FAM_WEIGHT_PACK, 10, 0.0, 8.0 * (double)total, (hipStream_t)stream);
  hipLaunchKernelGGL(upmix_split_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     wlo, ws, Co, Cu, Cs);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_upmix_merge_wgrad(const float* dwlo, const float* dws, float* dw, int Co, int Cu, int Cs,
                                      dram_stream_t stream) {
  if (!dwlo || !dws || !dw || Co < 1 || Cu < 1 || Cs < 1) return DRAM_ERR_BAD_ARG;
  const long total = (long)Co * (Cu + Cs) * 27;
  DramProf prof(DRAM_FAM_WEIGHT_PACK, 11, 0.0, 8.0 * (double)total, (hipStream_t)stream);
  hipLaunchKernelGGL(upmix_merge_wgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     dwlo, dws, dw, Co, Cu, Cs);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
