// stem.hip -- Conv3d(1, 64, k=7, s=2, p=3, bias=False): forward and weight gradient.
// Replaces nn.Conv3d at reference med3d.py:196-202 / :296-302 (fwd :272 / :371) and its
// autograd weight gradient.  (No data gradient: the network input needs none.)
//
// C_in = 1, so there is no channel row to gather: the im2col operand is read straight
// out of an LDS-resident input patch.  Forward: workgroup = 4x8x8 output voxels x 64
// channels; LDS holds the 13x21x21 input patch once and the weights of one kz-plane at
// a time (49 taps padded to 50 = 25 MFMA k-pairs); A[i=voxel][k=tap] is a ds_read_b32
// at patch[base(voxel) + off(tap)], B[k=tap][j=co] a conflict-free row read.
// Weight gradient: M = co (64), N = taps (343 -> 11 tiles of 32), K = voxels; a
// persistent workgroup walks 1x8x8 sub-tiles, keeps its 11 x 2 accumulators spread
// over 4 waves, and writes one slab; an ordered reduce produces [64][1][7][7][7].
#include "common.h"

namespace {

constexpr int PZ = 13, PY = 21, PX = 21;          // forward patch (4x8x8 outputs)
constexpr int PATCH = PZ * PY * PX;               // 5733 logical values
// LDS pitches chosen against bank conflicts (64 banks x 4 B; conflict cycles counted per wave-level ds_read_b32 for
// every lane pattern of the kernels): an x-row pitch of 24 puts the four output rows of a half-wave (offsets 2*24*ty
// = 0, 48, 32, 16 mod 64) on disjoint even banks -- 1.16 cycles per A read where the dense pitch 21 took 2.04 --
// and a weight-row pitch of 96 puts the two k rows of an MFMA (lanes 0-31 / 32-63) 32 banks apart (1 cycle, was 2).
constexpr int PXP = 24;                           // patch x-row pitch in LDS
constexpr int PATCHP = PZ * PY * PXP;             // 6552 floats
constexpr int WLP = 96;                           // weight row pitch in LDS
constexpr int KT = 50;                            // taps per kz plane, padded
constexpr int WPT = (KT * 64 + 255) / 256;        // weight floats per thread (13)

struct StemGeom {
  int B, D, H, W, Do, Ho, Wo;
  int nz, ny, nx, tiles_per_b, nblk;
};

template <typename T>   // T: storage type of the output y (float | bf16_t); the BN sums are taken from the fp32 accumulators
__global__ __launch_bounds__(256, 2) void stem_fwd_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ w,
                                                          T* __restrict__ y, float* __restrict__ stats,
                                                          const StemGeom g) {
  __shared__ __attribute__((aligned(16))) float lds[PATCHP + KT * WLP];
  float* patch = lds;
  float* wl = lds + PATCHP;  // [KT][WLP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;

  int L = xcd_remap(blockIdx.x, g.nblk);
  const int b = L / g.tiles_per_b;
  int r = L - b * g.tiles_per_b;
  const int txi = r % g.nx; r /= g.nx;
  const int tyi = r % g.ny;
  const int tzi = r / g.ny;
  const int z0 = tzi * 4, y0 = tyi * 8, x0 = txi * 8;

  // input patch (zero padded)
  for (int idx = tid; idx < PATCH; idx += 256) {
    const int pz = idx / (PY * PX);
    const int rem = idx - pz * (PY * PX);
    const int py = rem / PX, px = rem - py * PX;
    const int zi = 2 * z0 - 3 + pz, yi = 2 * y0 - 3 + py, xi = 2 * x0 - 3 + px;
    const bool ok = (zi >= 0) & (zi < g.D) & (yi >= 0) & (yi < g.H) & (xi >= 0) & (xi < g.W);
    patch[(pz * PY + py) * PXP + px] = ok ? x[(((long)b * g.D + zi) * g.H + yi) * g.W + xi] : 0.f;
  }

  float rw[WPT];
  auto load_w = [&](int kz) {
#pragma unroll
    for (int p = 0; p < WPT; ++p) {
      const int idx = p * 256 + tid;
      const int t = idx >> 6, co = idx & 63;
      rw[p] = (idx < KT * 64 && t < 49) ? w[co * 343 + kz * 49 + t] : 0.f;
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int p = 0; p < WPT; ++p) {
      const int idx = p * 256 + tid;
      if (idx < KT * 64) wl[(idx >> 6) * WLP + (idx & 63)] = rw[p];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  // rows of this wave: row = 64*wave + 32*mi + li -> (tz, ty, tx) = (wave, 4*mi + li>>3, li&7)
  int abase[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) abase[mi] = (2 * wave) * (PY * PXP) + (2 * (4 * mi + (li >> 3))) * PXP + 2 * (li & 7);

  load_w(0);
  for (int kz = 0; kz < 7; ++kz) {
    __syncthreads();
    store_w();
    __syncthreads();
    if (kz + 1 < 7) load_w(kz + 1);
    const float* pk = patch + kz * (PY * PXP);
#pragma unroll
    for (int kk = 0; kk < KT / 2; ++kk) {
      constexpr int dummy = 0;
      (void)dummy;
      const int k0 = 2 * kk, k1 = 2 * kk + 1;
      const int o0 = (k0 < 49) ? (k0 / 7) * PXP + (k0 % 7) : 0;
      const int o1 = (k1 < 49) ? (k1 / 7) * PXP + (k1 % 7) : 0;
      const int ko = lh ? o1 : o0;
      const float a0 = pk[abase[0] + ko];
      const float a1 = pk[abase[1] + ko];
      const float b0 = wl[(2 * kk + lh) * WLP + li];
      const float b1 = wl[(2 * kk + lh) * WLP + 32 + li];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
  }

  float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wave * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int zo = z0 + (row >> 6), yo = y0 + ((row >> 3) & 7), xo = x0 + (row & 7);
      if ((zo < g.Do) & (yo < g.Ho) & (xo < g.Wo)) {
        const long o = ((((long)b * g.Do + zo) * g.Ho + yo) * g.Wo + xo) * 64 + li;
#pragma unroll
        for (int nj = 0; nj < 2; ++nj) {
          const float v = acc[mi][nj][e];
          st1<T>(y, o + nj * 32, v);
          s1[nj] += v;
          s2[nj] += v * v;
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    float* red = lds;  // [4][2][64]
#pragma unroll
    for (int nj = 0; nj < 2; ++nj) {
      const float t1 = s1[nj] + __shfl_xor(s1[nj], 32, 64);
      const float t2 = s2[nj] + __shfl_xor(s2[nj], 32, 64);
      if (lh == 0) {
        red[(wave * 2 + 0) * 64 + nj * 32 + li] = t1;
        red[(wave * 2 + 1) * 64 + nj * 32 + li] = t2;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, c = tid & 63;
      const float v = red[(0 + which) * 64 + c] + red[(2 + which) * 64 + c] + red[(4 + which) * 64 + c] +
                      red[(6 + which) * 64 + c];
      stats[((long)L * 2 + which) * 64 + c] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------
constexpr int GZ = 7, GY = 21, GX = 21;   // wgrad patch: 1x8x8 outputs
constexpr int GPATCH = GZ * GY * GX;      // 3087 logical values
constexpr int GXP = 27;                   // x-row pitch in LDS: the 32 taps of a B read (7-value runs 27 apart, planes
                                          // 567 apart) then hit 1.18 bank cycles per read instead of 2.0 (pitch 21)
constexpr int GPATCHP = GZ * GY * GXP;    // 3969 floats
constexpr int NTAP = 352;                 // 343 padded to 11 x 32
constexpr int GLDY = 96;                  // dy row pitch: the two k rows of an MFMA 32 banks apart (was 68: 2-way)
constexpr int GPT = (GPATCH + 255) / 256; // 13

struct StemWGeom {
  int B, D, H, W, Do, Ho, Wo;
  int ny, nx;      // sub-tiles per (b, zo) plane
  int total;       // B*Do*ny*nx
};

template <typename T>   // T: storage type of dy
__global__ __launch_bounds__(256, 2) void stem_wgrad_kernel(const float* __restrict__ x,
                                                            const T* __restrict__ dy,
                                                            float* __restrict__ slab, const StemWGeom g) {
  __shared__ __attribute__((aligned(16))) float lds[GPATCHP + 3 + 64 * GLDY];
  float* patch = lds;
  float* dyl = lds + GPATCHP + 3;  // [64 vox][GLDY], 16-byte aligned (3972 floats in)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int ntw = (wave < 3) ? 3 : 2;  // N tiles of this wave: wave, wave+4, wave+8

  int toff[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tap = 32 * (wave + 4 * j) + li;
    const int kz = tap / 49, rem = tap - kz * 49, ky = rem / 7, kx = rem - ky * 7;
    toff[j] = (tap < 343 ? kz * (GY * GXP) + ky * GXP + kx : 0) + 2 * lh;
  }

  f32x16 acc[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][mi][e] = 0.f;

  float rp[GPT];
  float4 rd[4];
  auto load_tile = [&](int st) {
    int r = st;
    const int txi = r % g.nx; r /= g.nx;
    const int tyi = r % g.ny; r /= g.ny;
    const int zo = r % g.Do;
    const int b = r / g.Do;
    const int y0 = tyi * 8, x0 = txi * 8;
#pragma unroll
    for (int p = 0; p < GPT; ++p) {
      const int idx = p * 256 + tid;
      const int pz = idx / (GY * GX);
      const int rem = idx - pz * (GY * GX);
      const int py = rem / GX, px = rem - py * GX;
      const int zi = 2 * zo - 3 + pz, yi = 2 * y0 - 3 + py, xi = 2 * x0 - 3 + px;
      const bool ok = (idx < GPATCH) & (zi >= 0) & (zi < g.D) & (yi >= 0) & (yi < g.H) & (xi >= 0) & (xi < g.W);
      rp[p] = ok ? x[(((long)b * g.D + zi) * g.H + yi) * g.W + xi] : 0.f;
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = p * 256 + tid;  // 64 vox x 16 float4
      const int v = idx >> 4, c4 = idx & 15;
      const int yo = y0 + (v >> 3), xo = x0 + (v & 7);
      const bool ok = (yo < g.Ho) & (xo < g.Wo);
      const long o = ((((long)b * g.Do + zo) * g.Ho + yo) * g.Wo + xo) * 64 + c4 * 4;
      rd[p] = ok ? ld4<T>(dy, o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int p = 0; p < GPT; ++p) {
      const int idx = p * 256 + tid;
      if (idx < GPATCH) {
        const int pz = idx / (GY * GX), rem = idx - pz * (GY * GX);
        patch[(pz * GY + rem / GX) * GXP + rem % GX] = rp[p];
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int idx = p * 256 + tid;
      *reinterpret_cast<float4*>(&dyl[(idx >> 4) * GLDY + (idx & 15) * 4]) = rd[p];
    }
  };

  int st = blockIdx.x;
  if (st < g.total) load_tile(st);
  for (; st < g.total; st += gridDim.x) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (st + (int)gridDim.x < g.total) load_tile(st + gridDim.x);
#pragma unroll 8
    for (int kk = 0; kk < 32; ++kk) {
      // vox = 2*kk + lh -> (ty, tx) = (kk>>2, 2*(kk&3) + lh); the 2*lh is folded into toff
      const int vbase = (2 * (kk >> 2)) * GXP + 4 * (kk & 3);
      const float a0 = dyl[(2 * kk + lh) * GLDY + li];
      const float a1 = dyl[(2 * kk + lh) * GLDY + 32 + li];
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (j < ntw) {
          const float bv = patch[vbase + toff[j]];
          acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, bv, acc[j][0], 0, 0, 0);
          acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, bv, acc[j][1], 0, 0, 0);
        }
      }
    }
  }
  // slab[blk][co][NTAP]
  float* sl = slab + (long)blockIdx.x * 64 * NTAP;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < ntw) {
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = 32 * mi + (e & 3) + 8 * (e >> 2) + 4 * lh;
          sl[co * NTAP + 32 * (wave + 4 * j) + li] = acc[j][mi][e];
        }
    }
  }
}

// 64 results x 4 slab groups per block; a thread sums its group's slabs (k = kg, kg + 4, ...) in eight interleaved
// partial sums, the four groups are combined through LDS in a fixed order (deterministic).  One thread per result with
// a serial sum over 512 slabs was latency-bound: 120 us.
__global__ __launch_bounds__(256) void stem_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                int nslab) {
  __shared__ float part[4][64];
  const int r = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + r;  // over 64*NTAP
  const bool live = i < 64 * NTAP;
  float p[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) p[j] = 0.f;
  if (live) {
    int k = kg;
    for (; k + 28 < nslab; k += 32) {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] += slab[(long)(k + 4 * j) * 64 * NTAP + i];
    }
    for (int j = 0; k < nslab; k += 4, ++j) p[j] += slab[(long)k * 64 * NTAP + i];
  }
  part[kg][r] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  __syncthreads();
  if (kg == 0 && live) {
    const int co = i / NTAP, tap = i - co * NTAP;
    if (tap < 343) dw[co * 343 + tap] = (part[0][r] + part[1][r]) + (part[2][r] + part[3][r]);
  }
}

inline int stem_out(int n) { return (n + 6 - 7) / 2 + 1; }
inline int stem_wgrad_blocks(int total) { return total < 512 ? total : 512; }

}  // namespace

extern "C" int dram_stem_num_tiles(int B, int Do, int Ho, int Wo) {
  return B * ((Do + 3) / 4) * ((Ho + 7) / 8) * ((Wo + 7) / 8);
}

template <typename T>
static int stem_fwd_impl(const float* x, const float* w, T* y, float* stats_partial, int B, int D, int H, int W,
                         dram_stream_t stream) {
  if (!x || !w || !y || B < 1 || D < 1 || H < 1 || W < 1) return DRAM_ERR_BAD_ARG;
  StemGeom g{};
  g.B = B; g.D = D; g.H = H; g.W = W;
  g.Do = stem_out(D); g.Ho = stem_out(H); g.Wo = stem_out(W);
  if ((long long)B * g.Do * g.Ho * g.Wo * 64 >= (1LL << 31)) return DRAM_ERR_UNSUPPORTED;
  g.nz = (g.Do + 3) / 4; g.ny = (g.Ho + 7) / 8; g.nx = (g.Wo + 7) / 8;
  g.tiles_per_b = g.nz * g.ny * g.nx;
  g.nblk = B * g.tiles_per_b;
  const double vo = (double)B * g.Do * g.Ho * g.Wo;
  DramProf prof(DRAM_FAM_STEM, 0, 2.0 * vo * 64.0 * 343.0,
                4.0 * ((double)B * D * H * W + 64.0 * 343.0) + sizeof(T) * vo * 64.0, (hipStream_t)stream);
  hipLaunchKernelGGL((stem_fwd_kernel<T>), dim3(g.nblk), dim3(256), 0, (hipStream_t)stream, x, w, y, stats_partial, g);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_stem_fwd(const float* x, const float* w, float* y, float* stats_partial, int B, int D,
                             int H, int W, dram_stream_t stream) {
  return stem_fwd_impl<float>(x, w, y, stats_partial, B, D, H, W, stream);
}
extern "C" int dram_stem_fwd_bf16(const float* x, const float* w, void* y, float* stats_partial, int B, int D,
                                  int H, int W, dram_stream_t stream) {
  return stem_fwd_impl<bf16_t>(x, w, (bf16_t*)y, stats_partial, B, D, H, W, stream);
}

extern "C" size_t dram_stem_bwd_weight_workspace(int B, int D, int H, int W) {
  const int Do = stem_out(D), Ho = stem_out(H), Wo = stem_out(W);
  const int total = B * Do * ((Ho + 7) / 8) * ((Wo + 7) / 8);
  return (size_t)stem_wgrad_blocks(total) * 64 * NTAP * sizeof(float);
}

template <typename T>
static int stem_bwd_weight_impl(const float* x, const T* dy, float* dw, int B, int D, int H, int W, void* workspace,
                                size_t workspace_bytes, dram_stream_t stream) {
  if (!x || !dy || !dw || B < 1 || D < 1 || H < 1 || W < 1) return DRAM_ERR_BAD_ARG;
  StemWGeom g{};
  g.B = B; g.D = D; g.H = H; g.W = W;
  g.Do = stem_out(D); g.Ho = stem_out(H); g.Wo = stem_out(W);
  g.ny = (g.Ho + 7) / 8; g.nx = (g.Wo + 7) / 8;
  g.total = B * g.Do * g.ny * g.nx;
  const int nblk = stem_wgrad_blocks(g.total);
  if (!workspace || workspace_bytes < (size_t)nblk * 64 * NTAP * sizeof(float)) return DRAM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const double vo = (double)B * g.Do * g.Ho * g.Wo;
  DramProf prof(DRAM_FAM_STEM, 1, 2.0 * vo * 64.0 * 343.0,
                4.0 * ((double)B * D * H * W + 64.0 * 343.0) + sizeof(T) * vo * 64.0, s);
  hipLaunchKernelGGL((stem_wgrad_kernel<T>), dim3(nblk), dim3(256), 0, s, x, dy, (float*)workspace, g);
  DRAM_LAUNCH_CHECK();
  hipLaunchKernelGGL(stem_wgrad_reduce_kernel, dim3((64 * NTAP + 63) / 64), dim3(256), 0, s,
                     (const float*)workspace, dw, nblk);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_stem_bwd_weight(const float* x, const float* dy, float* dw, int B, int D, int H, int W,
                                    void* workspace, size_t workspace_bytes, dram_stream_t stream) {
  return stem_bwd_weight_impl<float>(x, dy, dw, B, D, H, W, workspace, workspace_bytes, stream);
}
extern "C" int dram_stem_bwd_weight_bf16(const float* x, const void* dy, float* dw, int B, int D, int H, int W,
                                         void* workspace, size_t workspace_bytes, dram_stream_t stream) {
  return stem_bwd_weight_impl<bf16_t>(x, (const bf16_t*)dy, dw, B, D, H, W, workspace, workspace_bytes, stream);
}
