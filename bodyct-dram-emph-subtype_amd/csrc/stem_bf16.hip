// stem_bf16.hip -- Conv3d(1, 64, k=7, s=2, p=3, bias=False) on the bf16 matrix cores (bf16 storage path): forward
// and weight gradient.  Replaces the fp32-MFMA stem (stem.hip) when the activations are bf16 (reference:
// `--precision bf16`, train.py:46: autocast runs this convolution in bf16 too).  The input volume stays fp32 in HBM
// (C_in = 1, the caller's tensor) and is rounded to bf16 on its way into LDS; products are exact in fp32 accumulators.
//
// C_in = 1: the GEMM K dimension is the 343 taps, and an MFMA k16 step must be 16 taps whose input values sit in
// CONSECUTIVE patch elements.  Forward: k16 = (two ky rows) x (kx = 0..7, the eighth a zero weight): lane (voxel, h)
// reads the 8 consecutive bf16 patch values of row 2y + 2 kyp + h starting at 2x -- four ds_read_b32 -- and the
// weights live in LDS pre-arranged as [kz][kyp][h][co][8].  28 MFMA steps per 32x32 block instead of 172 fp32 ones.
// Weight gradient: M = co, N = taps (343 -> 11 blocks of 32), K = output voxels; the A operand is a transposed read
// of the dy tile (ds_read_b64_tr_b16), the B operand needs x[2 v + tap] for 8 consecutive v: stride 2 in the patch,
// so the patch is stored de-interleaved in four images per row (even / odd x phase, each also shifted by one), which
// makes every tap's run 8 consecutive, 4-byte aligned bf16 values.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

constexpr int BUF_FLAGS = 0x00020000;
#define BUFLDS16(rsrc_, voff_, dst_)                                                                               \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc_), (__attribute__((address_space(3))) void*)(dst_), 16, (voff_), 0, 0, 0)

inline int stem_out(int n) { return (n + 6 - 7) / 2 + 1; }

struct SGeom {
  int B, D, H, W, Do, Ho, Wo;
  int nz, ny, nx, tiles_per_b, ntile;
};

// ---------------------------------------------------------------------------------------------------------
// forward: persistent workgroups (the 56 KB weight image is built once), tile = 4 x 8 x 8 output voxels x 64 channels
constexpr int FPZ = 13, FPY = 22, FPX = 24;           // patch: 13 planes x 22 rows (21 + a zero row) x 24 (21 + zeros)
constexpr int FPATCH = FPZ * FPY * FPX;               // 6864 bf16
constexpr int FLOADS = (13 * 21 * 21 + 255) / 256;    // 23 values per thread

__global__ __launch_bounds__(256, 2) void stem_bf16_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               bf16_t* __restrict__ y, float* __restrict__ stats,
                                                               const SGeom g) {
  __shared__ __attribute__((aligned(16))) unsigned char wl[7 * 4 * 2 * 64 * 16];      // [kz][kyp][h][co][8 bf16]
  __shared__ __attribute__((aligned(16))) bf16_t patch[FPATCH];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;

  // weights: fp32 [co][kz][ky][kx] -> bf16 image, ky = 7 and kx = 7 are zero columns
  for (int i = tid; i < 7 * 8 * 64 * 8; i += 256) {
    const int kx = i & 7, co = (i >> 3) & 63, ky = (i >> 9) & 7, kz = i >> 12;
    const float v = (ky < 7 && kx < 7) ? w[co * 343 + kz * 49 + ky * 7 + kx] : 0.f;
    reinterpret_cast<bf16_t*>(wl)[((((kz * 4 + (ky >> 1)) * 2 + (ky & 1)) * 64 + co) << 3) + kx] = f32_to_bf16(v);
  }
  for (int i = tid; i < FPATCH; i += 256) patch[i] = 0;        // the pad row / columns stay zero for good

  float rp[FLOADS];
  auto load_patch = [&](int t) __attribute__((always_inline)) {
    const int b = t / g.tiles_per_b;
    int r = t - b * g.tiles_per_b;
    const int txi = r % g.nx; r /= g.nx;
    const int tyi = r % g.ny;
    const int tzi = r / g.ny;
#pragma unroll
    for (int p = 0; p < FLOADS; ++p) {
      const int idx = p * 256 + tid;
      const int pz = idx / (21 * 21), rem = idx - pz * (21 * 21), py = rem / 21, px = rem - py * 21;
      const int zi = 8 * tzi - 3 + pz, yi = 16 * tyi - 3 + py, xi = 16 * txi - 3 + px;
      const bool ok = (idx < 13 * 21 * 21) & (zi >= 0) & (zi < g.D) & (yi >= 0) & (yi < g.H) & (xi >= 0) & (xi < g.W);
      rp[p] = ok ? x[(((long)b * g.D + zi) * g.H + yi) * g.W + xi] : 0.f;
    }
  };
  auto store_patch = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < FLOADS; ++p) {
      const int idx = p * 256 + tid;
      if (idx < 13 * 21 * 21) {
        const int pz = idx / (21 * 21), rem = idx - pz * (21 * 21), py = rem / 21, px = rem - py * 21;
        patch[(pz * FPY + py) * FPX + px] = f32_to_bf16(rp[p]);
      }
    }
  };

  // A row li of block mi: output voxel (z = wave, y = 4 mi + li / 8, x = li % 8) -> patch element (2z + kz, 2y + ky, 2x)
  int abase[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) abase[mi] = ((2 * wave) * FPY + 2 * (4 * mi + (li >> 3)) + lh) * FPX + 2 * (li & 7);
  const int bbase = (lh * 64 + li) * 16;

  int t = blockIdx.x;
  if (t < g.ntile) load_patch(t);
  for (; t < g.ntile; t += gridDim.x) {
    __syncthreads();                           // the last tile's MFMAs are done with the patch (first trip: the fills above)
    store_patch();
    __syncthreads();
    if (t + (int)gridDim.x < g.ntile) load_patch(t + gridDim.x);

    f32x16 acc[2][2];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[mi][nb][e] = 0.f;
#pragma unroll 1
    for (int kz = 0; kz < 7; ++kz) {
#pragma unroll
      for (int kyp = 0; kyp < 4; ++kyp) {
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) {
          const unsigned* pa = reinterpret_cast<const unsigned*>(patch + abase[mi] + (kz * FPY + 2 * kyp) * FPX);
          u32x4 v;
          v[0] = pa[0]; v[1] = pa[1]; v[2] = pa[2]; v[3] = pa[3];
          af[mi] = __builtin_bit_cast(bf16x8, v);
        }
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
          bfr[nb] = *reinterpret_cast<const bf16x8*>(wl + (kz * 4 + kyp) * 2048 + bbase + nb * 512);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int nb = 0; nb < 2; ++nb)
            acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[nb], acc[mi][nb], 0, 0, 0);
      }
    }

    const int b = t / g.tiles_per_b;
    int r = t - b * g.tiles_per_b;
    const int txi = r % g.nx; r /= g.nx;
    const int tyi = r % g.ny;
    const int tzi = r / g.ny;
    const int zo = 4 * tzi + wave;
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f};
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        const int yo = 8 * tyi + 4 * mi + (row >> 3), xo = 8 * txi + (row & 7);
        if ((zo < g.Do) & (yo < g.Ho) & (xo < g.Wo)) {
          const long o = ((((long)b * g.Do + zo) * g.Ho + yo) * g.Wo + xo) * 64 + li;
#pragma unroll
          for (int nb = 0; nb < 2; ++nb) {
            const bf16_t hv = f32_to_bf16(acc[mi][nb][e]);
            y[o + nb * 32] = hv;
            const float vr = bf16_to_f32(hv);
            s1[nb] += vr;
            s2[nb] += vr * vr;
          }
        }
      }
    if (stats) {
      __shared__ float red[4][2][64];
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        const float t1 = s1[nb] + __shfl_xor(s1[nb], 32, 64);
        const float t2 = s2[nb] + __shfl_xor(s2[nb], 32, 64);
        if (lh == 0) {
          red[wave][0][nb * 32 + li] = t1;
          red[wave][1][nb * 32 + li] = t2;
        }
      }
      __syncthreads();
      if (tid < 128) {
        const int which = tid >> 6, cc = tid & 63;
        stats[((long)t * 2 + which) * 64 + cc] = red[0][which][cc] + red[1][which][cc] + red[2][which][cc] + red[3][which][cc];
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient: persistent workgroups walk 1 x 8 x 8 sub-tiles; slab[blk][co][352] (the layout of stem.hip's
// slabs: its ordered reduce finishes the job)
constexpr int NTAP = 352;
constexpr int GPY = 22, GROW = 48;                    // rows per plane (21 + zero row), elements per row: 4 images x 12
constexpr int GPATCH = 7 * GPY * GROW;                // 7392 bf16
constexpr int GLOADS = (7 * 21 * 21 + 255) / 256;     // 13

struct SWGeom {
  int B, D, H, W, Do, Ho, Wo;
  int ny, nx, total;
};

__global__ __launch_bounds__(256, 2) void stem_bf16_wgrad_kernel(const float* __restrict__ x, const bf16_t* __restrict__ dy,
                                                                 float* __restrict__ slab, const SWGeom g) {
  __shared__ __attribute__((aligned(16))) bf16_t patch[GPATCH];        // [pz][row][pe 12 | pe1 12 | po 12 | po1 12]
  __shared__ __attribute__((aligned(1024))) unsigned char dyt[8192];    // [2 cb][64 voxels][64 B]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ntw = (wave < 3) ? 3 : 2;                 // tap blocks of this wave: wave, wave + 4, wave + 8
  for (int i = tid; i < GPATCH; i += 256) patch[i] = 0;

  // B operand: lane = tap t = 32 (wave + 4 j) + li, k half lh: voxels (y = 2 s + lh, x = 0..7) of k16 step s
  // -> patch (kz, 2 y + ky) elements x' = 2 x + kx: phase image (kx & 1), start index m = kx >> 1; an odd m reads
  // the image shifted by one at m - 1 (4-byte aligned)
  int toff[3];
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int tap = 32 * (wave + 4 * j) + li;
    const int kz = tap / 49, rem = tap - kz * 49, ky = rem / 7, kx = rem - ky * 7;
    const int m = kx >> 1, img = (kx & 1) * 2 + (m & 1);
    toff[j] = tap < 343 ? ((kz * GPY + ky + 2 * lh) * GROW + img * 12 + (m & ~1)) : 0;
  }
  // A operand (dy^T): transposed reads, rows = voxels (2 s + h, 4 r + q), see wgrad3_bf16_kernel
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3, h = g4 >> 1;
  const int cpiece = ((g4 & 1) * 2 + (p4 >> 1)) * 16 + (p4 & 1) * 8;
  const int a0 = (h * 8 + q) * 64 + cpiece, a1 = (h * 8 + 4 + q) * 64 + cpiece;
  auto tr8 = [&](const unsigned char* base) __attribute__((always_inline)) {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a0));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + a1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  f32x16 acc[3][2];
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][cb][e] = 0.f;

  const long dybytes = (long)g.B * g.Do * g.Ho * g.Wo * 64 * 2;
  float rp[GLOADS];
  auto load_tile = [&](int st) __attribute__((always_inline)) {
    int r = st;
    const int txi = r % g.nx; r /= g.nx;
    const int tyi = r % g.ny; r /= g.ny;
    const int zo = r % g.Do;
    const int b = r / g.Do;
#pragma unroll
    for (int p = 0; p < GLOADS; ++p) {
      const int idx = p * 256 + tid;
      const int pz = idx / (21 * 21), rem = idx - pz * (21 * 21), py = rem / 21, px = rem - py * 21;
      const int zi = 2 * zo - 3 + pz, yi = 16 * tyi - 3 + py, xi = 16 * txi - 3 + px;
      const bool ok = (idx < 7 * 21 * 21) & (zi >= 0) & (zi < g.D) & (yi >= 0) & (yi < g.H) & (xi >= 0) & (xi < g.W);
      rp[p] = ok ? x[(((long)b * g.D + zi) * g.H + yi) * g.W + xi] : 0.f;
    }
  };
  auto store_tile = [&](int st) __attribute__((always_inline)) {
    int r = st;
    const int txi = r % g.nx; r /= g.nx;
    const int tyi = r % g.ny; r /= g.ny;
    const int zo = r % g.Do;
    const int b = r / g.Do;
#pragma unroll
    for (int p = 0; p < GLOADS; ++p) {
      const int idx = p * 256 + tid;
      if (idx < 7 * 21 * 21) {
        const int pz = idx / (21 * 21), rem = idx - pz * (21 * 21), py = rem / 21, px = rem - py * 21;
        const bf16_t v = f32_to_bf16(rp[p]);
        bf16_t* row = patch + (pz * GPY + py) * GROW + (px & 1) * 24;      // phase image pe | po
        const int i = px >> 1;
        row[i] = v;
        if (i >= 1) row[12 + i - 1] = v;                                   // the image shifted by one
      }
    }
    // dy tile by LDS-DMA: granule p -> (cb, voxel, slot); out-of-range voxels read zeros (empty offset)
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(dy)), 0, (int)(dybytes > 0xffffffffL ? 0xffffffffL : dybytes), BUF_FLAGS);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int p = i * 256 + tid, cb = p >> 8, v = (p >> 2) & 63, sl = p & 3;
      const int yo = 8 * tyi + (v >> 3), xo = 8 * txi + (v & 7);
      const bool ok = (yo < g.Ho) & (xo < g.Wo);
      const long e = ((((long)b * g.Do + zo) * g.Ho + yo) * g.Wo + xo) * 64 + cb * 32 + sl * 8;
      BUFLDS16(rs, ok ? (unsigned)(e * 2) : 0xffffffffu, dyt + i * 4096 + wave * 1024);
    }
  };

  int st = blockIdx.x;
  if (st < g.total) load_tile(st);
  for (; st < g.total; st += gridDim.x) {
    __syncthreads();                                  // the last sub-tile's reads are done
    store_tile(st);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (st + (int)gridDim.x < g.total) load_tile(st + gridDim.x);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 d0 = tr8(dyt + s * 1024), d1 = tr8(dyt + 4096 + s * 1024);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        if (j < ntw) {
          const unsigned* pb = reinterpret_cast<const unsigned*>(patch + toff[j] + 4 * s * GROW);
          u32x4 v;
          v[0] = pb[0]; v[1] = pb[1]; v[2] = pb[2]; v[3] = pb[3];
          const bf16x8 bfr = __builtin_bit_cast(bf16x8, v);
          acc[j][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d0, bfr, acc[j][0], 0, 0, 0);
          acc[j][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(d1, bfr, acc[j][1], 0, 0, 0);
        }
      }
    }
  }
  float* sl = slab + (long)blockIdx.x * 64 * NTAP;
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    if (j < ntw) {
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int co = 32 * cb + (e & 3) + 8 * (e >> 2) + 4 * lh;
          sl[co * NTAP + 32 * (wave + 4 * j) + li] = acc[j][cb][e];
        }
    }
  }
}

// 64 results x 4 slab groups per block; a thread sums its group's slabs (k = kg, kg + 4, ...) in eight interleaved
// partial sums, the four groups are combined through LDS in a fixed order (deterministic).  One thread per result with
// a serial sum over 512 slabs was latency-bound: 119 us.
__global__ __launch_bounds__(256) void stem_bf16_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                     int nslab) {
  __shared__ float part[4][64];
  const int r = threadIdx.x & 63, kg = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + r;  // over 64 * NTAP
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

inline int wgrad_blocks(int total) { return total < 512 ? total : 512; }

}  // namespace

// bf16-MFMA forms of dram_stem_fwd_bf16 / dram_stem_bwd_weight_bf16 (stem.hip keeps the fp32-MFMA forms with a bf16
// store / load; DRAM_STEM_BF16=0 selects those: A/B, tests)
extern "C" int dram_stem_fwd_bf16mm(const float* x, const float* w, void* y, float* stats_partial, int B, int D, int H,
                                    int W, dram_stream_t stream) {
  if (!x || !w || !y || B < 1 || D < 1 || H < 1 || W < 1) return DRAM_ERR_BAD_ARG;
  SGeom g{};
  g.B = B; g.D = D; g.H = H; g.W = W;
  g.Do = stem_out(D); g.Ho = stem_out(H); g.Wo = stem_out(W);
  g.nz = (g.Do + 3) / 4; g.ny = (g.Ho + 7) / 8; g.nx = (g.Wo + 7) / 8;
  g.tiles_per_b = g.nz * g.ny * g.nx;
  const long long nt = (long long)B * g.tiles_per_b;
  if (nt >= (1LL << 31)) return DRAM_ERR_UNSUPPORTED;
  g.ntile = (int)nt;
  const double vo = (double)B * g.Do * g.Ho * g.Wo;
  DramProf prof(DRAM_FAM_STEM, 2, 2.0 * (double)g.ntile * 256.0 * 64.0 * 448.0,
                4.0 * ((double)B * D * H * W + 64.0 * 343.0) + 2.0 * vo * 64.0, (hipStream_t)stream, 2.0 * vo * 64.0 * 343.0);
  hipLaunchKernelGGL(stem_bf16_fwd_kernel, dim3(g.ntile < 512 ? g.ntile : 512), dim3(256), 0, (hipStream_t)stream, x, w,
                     (bf16_t*)y, stats_partial, g);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_stem_bwd_weight_bf16mm(const float* x, const void* dy, float* dw, int B, int D, int H, int W,
                                           void* workspace, size_t workspace_bytes, dram_stream_t stream) {
  if (!x || !dy || !dw || B < 1 || D < 1 || H < 1 || W < 1) return DRAM_ERR_BAD_ARG;
  SWGeom g{};
  g.B = B; g.D = D; g.H = H; g.W = W;
  g.Do = stem_out(D); g.Ho = stem_out(H); g.Wo = stem_out(W);
  g.ny = (g.Ho + 7) / 8; g.nx = (g.Wo + 7) / 8;
  g.total = B * g.Do * g.ny * g.nx;
  const int nblk = wgrad_blocks(g.total);
  if (!workspace || workspace_bytes < (size_t)nblk * 64 * NTAP * sizeof(float)) return DRAM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const double vo = (double)B * g.Do * g.Ho * g.Wo;
  DramProf prof(DRAM_FAM_STEM, 3, 2.0 * (double)g.total * 64.0 * 64.0 * 352.0,
                4.0 * ((double)B * D * H * W + 64.0 * 343.0) + 2.0 * vo * 64.0, s, 2.0 * vo * 64.0 * 343.0);
  hipLaunchKernelGGL(stem_bf16_wgrad_kernel, dim3(nblk), dim3(256), 0, s, x, (const bf16_t*)dy, (float*)workspace, g);
  DRAM_LAUNCH_CHECK();
  hipLaunchKernelGGL(stem_bf16_wgrad_reduce_kernel, dim3((64 * NTAP + 63) / 64), dim3(256), 0, s,
                     (const float*)workspace, dw, nblk);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
