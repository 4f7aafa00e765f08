// conv_wgrad_w2d.hip -- weight gradient of the narrow stride-1 3x3x3 convolutions through the
// in-plane Winograd F(2x2, 3x3) domain, z-walking (companion of conv_wino2d.hip, which does the
// forward and the data gradient of the same layers).
//
// Replaces autograd's convolution_backward (weight) for the conv3x3x3 sites of reference
// med3d.py:91-100 (layer1) and :67/:76 (decoder) -- same role as conv_wgrad2_kernel.
//
//   dU2[xi][a][co][ci] = sum over (b, z, 2x2 tile t)  (A dy_t(z) A^T)[xi][co] * (B^T x_t(z+a-1) B)[xi][ci]
//   dw[co][ci][a][ky][kx] = (G^T dU2[.][a][co][ci] G)[ky][kx]
// 16 x 3 = 48 products per 2x2 tile and channel pair instead of 4 x 27 = 108 (2.25x fewer MFMAs).
//
// A workgroup of 8 waves owns a (32 co x 32 ci) block and walks columns of 8x8 voxels (4x4 tiles)
// through all planes, like conv_wgrad2_kernel.  LDS: a ring of 2 input planes (10x10 halo rows x
// 32 ci) and 2 dy planes (64 voxels x 32 co), filled by LDS-DMA one plane ahead.  Wave w owns the
// Winograd points xi = 2w, 2w+1 and all three z-taps: 6 accumulators.  Per plane it forms
//   * B^T x B of the NEW input plane z+1 for its two points (4 ds_read_b32 + 4 FMAs per value; the
//     values of planes z-1, z, z+1 stay in a register ring, so each is computed once and used by
//     three z-taps),
//   * A dy A^T of plane z (up to 4 reads),
// and issues 3 MFMAs (K = 2 tiles) per (point, tile pair).  Point-dependent positions and signs are
// wave-uniform scalars, so all waves run the same code.  Operand rows are [voxel][channel]: lanes
// are consecutive channels (conflict-free ds_read_b32, no swizzle).
// Split over columns into slabs; a second kernel sums the slabs in a fixed order and applies
// G^T . G -> deterministic, writes the reference layout [Cout][Cin][3][3][3].
#include <stdlib.h>
#include "common.h"

namespace {

__device__ __attribute__((aligned(128))) float g_w2g_zero_line[32];

struct WW2Geom {
  int B, D, H, W, Cin, Cout;
  int ny, nx, ncols;               // 8x8 columns per plane, total columns = B*ny*nx
  int ci_tiles, nslab, cpw;        // 32-wide ci tiles, workgroups per (co,ci) pair, columns per workgroup
};

// B^T rows (two non-zeros each): positions and signs;  A rows: coefficients of the two outputs
__constant__ int c_bp[4][2] = {{0, 2}, {1, 2}, {1, 2}, {1, 3}};
__constant__ float c_bs[4][2] = {{1.f, -1.f}, {1.f, 1.f}, {-1.f, 1.f}, {1.f, -1.f}};
__constant__ float c_a[4][2] = {{1.f, 0.f}, {1.f, 1.f}, {1.f, -1.f}, {0.f, -1.f}};

__global__ __launch_bounds__(512) void conv_wgrad_w2d_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                             float* __restrict__ slab, const WW2Geom g) {
  constexpr int PR = 104;                 // rows per halo plane slot (100 used, 13 DMA pieces of 8)
  constexpr int HPL = PR * 32;            // floats per halo plane
  constexpr int DPL = 64 * 32;            // floats per dy plane
  __shared__ __attribute__((aligned(1024))) float lds[2 * HPL + 2 * DPL];
  float* halo = lds;
  float* dyl = lds + 2 * HPL;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int sub = lane >> 3, pslot = lane & 7;

  const int pair = blockIdx.x / g.nslab, sidx = blockIdx.x - pair * g.nslab;
  const int co0 = (pair / g.ci_tiles) * 32, ci0 = (pair % g.ci_tiles) * 32;
  const int c_begin = sidx * g.cpw;
  const int c_end = (c_begin + g.cpw < g.ncols) ? c_begin + g.cpw : g.ncols;

  // ---- this wave's two Winograd points: operand positions (float offsets) and signs ------------
  int xo[2][4];      // halo offsets of the four raw positions (p, q) of B^T x B
  float xs[2][4];
  float dc[2][4];    // coefficients of dy[oy][ox] in A dy A^T  (offsets are fixed: (oy*8 + ox)*32)
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int xi = 2 * wave + j, xy = xi >> 2, xx = xi & 3;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        xo[j][u * 2 + v] = (c_bp[xy][u] * 10 + c_bp[xx][v]) * 32;
        xs[j][u * 2 + v] = c_bs[xy][u] * c_bs[xx][v];
        dc[j][u * 2 + v] = c_a[xy][u] * c_a[xx][v];
      }
  }
  // lane part of the operand addresses: tile t = 2*kk + lh -> (ty, tx) = (kk >> 1, 2*(kk & 1) + lh)
  const int xlane = lh * 2 * 32 + li;          // x halo: x_h = 2*tx + q -> the lh tile sits 2 voxels to the right
  const int dlane = lh * 2 * 32 + li;          // dy: x = 2*tx + ox, likewise

  f32x16 acc[2][3];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[j][a][e] = 0.f;

  const float* zline = g_w2g_zero_line + pslot * 4;
  const long plane_x = (long)g.H * g.W * g.Cin, plane_dy = (long)g.H * g.W * g.Cout;

  for (int col = c_begin; col < c_end; ++col) {
    int r = col;
    const int txi = r % g.nx; r /= g.nx;
    const int tyi = r % g.ny;
    const int b = r / g.ny;
    const int y0 = tyi * 8, x0 = txi * 8;
    long hsrc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int row = 8 * (wave + 8 * q) + sub;       // halo row 0..103
      const int yh = row / 10, xh = row - yh * 10;
      const int yi = y0 - 1 + yh, xi = x0 - 1 + xh;
      const bool v = (row < 100) & (yi >= 0) & (yi < g.H) & (xi >= 0) & (xi < g.W);
      hsrc[q] = v ? ((long)yi * g.W + xi) * g.Cin + ci0 + pslot * 4 : -1;
    }
    long dsrc;
    {
      const int vox = 8 * wave + sub;                 // dy row 0..63
      const int yo = y0 + (vox >> 3), xo_ = x0 + (vox & 7);
      dsrc = (yo < g.H && xo_ < g.W) ? ((long)yo * g.W + xo_) * g.Cout + co0 + pslot * 4 : -1;
    }
    const float* xb = x + (long)b * g.D * plane_x;
    const float* db = dy + (long)b * g.D * plane_dy;

    auto issue_halo = [&](int zi) __attribute__((always_inline)) {   // input plane zi -> slot zi & 1
      float* dst = halo + (zi & 1) * HPL;
      const bool zin = (zi >= 0) & (zi < g.D);
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        if (wave + 8 * q < 13) {
          const float* src = (zin && hsrc[q] >= 0) ? xb + (long)zi * plane_x + hsrc[q] : zline;
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                           (__attribute__((address_space(3))) void*)(dst + (wave + 8 * q) * 256), 16, 0, 0);
        }
      }
    };
    auto issue_dy = [&](int zo) __attribute__((always_inline)) {     // dy plane zo -> slot zo & 1
      const float* src = (zo >= 0 && zo < g.D && dsrc >= 0) ? db + (long)zo * plane_dy + dsrc : zline;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dyl + (zo & 1) * DPL + wave * 256), 16, 0, 0);
    };

    // register ring of the transformed input: xr[slot][point][tile pair], plane p lives in slot (p + 3) % 3
    float xr[3][2][8];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) xr[2][j][kk] = 0.f;          // plane -1: zero padding

    __syncthreads();                    // previous column fully consumed
    issue_halo(0);                      // step s = -1 needs input plane 0
    // step s: transform input plane s+1 (slot (s+1)&1); for s >= 0 also dy plane s and the MFMAs
    for (int s0 = -1; s0 < g.D; s0 += 3) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int s = s0 + u;
        if (s < g.D) {                  // uniform over the workgroup
          __syncthreads();              // input plane s+1 and dy plane s have landed; step s-1 is finished everywhere
          issue_halo(s + 2);            // slot (s+2)&1 held plane s  (already transformed at step s-1)
          issue_dy(s + 1);              // slot (s+1)&1 held dy plane s-1
          // slots of planes s-1, s, s+1 in the register ring: (s+2)%3, (s+3)%3, (s+4)%3 with s = s0+u, s0 = -1 (mod 3)
          constexpr int kSlot[3][3] = {{1, 2, 0}, {2, 0, 1}, {0, 1, 2}};   // [u][a]  for s0 % 3 == 2 (i.e. s0 = -1 + 3n)
          const float* hp = halo + ((s + 1) & 1) * HPL + xlane;
          const float* dp = dyl + (s & 1) * DPL + dlane;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
              // tile pair kk: ty = kk >> 1, tx = 2*(kk & 1) + lh;  halo origin (2*ty, 2*tx), dy origin the same
              const int hoff = ((2 * (kk >> 1)) * 10 + 4 * (kk & 1)) * 32;
              const int doff = ((2 * (kk >> 1)) * 8 + 4 * (kk & 1)) * 32;
              float xv = xs[j][0] * hp[hoff + xo[j][0]];
              xv = __builtin_fmaf(xs[j][1], hp[hoff + xo[j][1]], xv);
              xv = __builtin_fmaf(xs[j][2], hp[hoff + xo[j][2]], xv);
              xv = __builtin_fmaf(xs[j][3], hp[hoff + xo[j][3]], xv);
              xr[kSlot[u][2]][j][kk] = xv;
              if (s >= 0) {             // uniform
                float dv = dc[j][0] * dp[doff];
                dv = __builtin_fmaf(dc[j][1], dp[doff + 32], dv);
                dv = __builtin_fmaf(dc[j][2], dp[doff + 8 * 32], dv);
                dv = __builtin_fmaf(dc[j][3], dp[doff + 9 * 32], dv);
#pragma unroll
                for (int a = 0; a < 3; ++a)
                  acc[j][a] = __builtin_amdgcn_mfma_f32_32x32x2f32(dv, xr[kSlot[u][a]][j][kk], acc[j][a], 0, 0, 0);
              }
            }
          }
        }
      }
    }
  }

  // ---- write this workgroup's slab slice: slab[sidx][xi*3 + a][co][ci] ---------------------------
  const long tap_stride = (long)g.Cout * g.Cin;
  float* sl = slab + (long)sidx * 48 * tap_stride;
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      const int t = (2 * wave + j) * 3 + a;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        sl[t * tap_stride + (long)co * g.Cin + ci0 + li] = acc[j][a][e];
      }
    }
}

// dw[co][ci][a][ky][kx] = (G^T (sum_s slab[s][.][a][co][ci]) G)[ky][kx],  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
__device__ __forceinline__ void gt3w(const float a0, const float a1, const float a2, const float a3, float* r) {
  r[0] = a0 + 0.5f * (a1 + a2);
  r[1] = 0.5f * (a1 - a2);
  r[2] = 0.5f * (a1 + a2) + a3;
}

// One workgroup = 64 (co, ci) elements x 16 Winograd points (1024 threads): every thread sums one
// (point, element) over the slabs (coalesced over elements; hundreds of slabs on the big layers, so
// the parallelism has to come from every point), then 64 threads apply G^T . G.
__global__ __launch_bounds__(1024) void wgrad_w2d_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                                const int nslab, const int Cout, const int Cin) {
  __shared__ float us[16][64];
  const long n = (long)Cout * Cin;
  const int e = threadIdx.x & 63, xi = threadIdx.x >> 6;
  const long i = blockIdx.x * 64L + e;               // (co, ci), ci fastest
  const int a = blockIdx.y;
  float s = 0.f;
  if (i < n) {
    // eight interleaved partial sums (eight loads in flight; a serial sum was latency-bound), fixed combination order
    const float* src = slab + ((long)xi * 3 + a) * n + i;
    float p[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = 0.f;
    int k = 0;
    for (; k + 8 <= nslab; k += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] += src[(long)(k + j) * 48 * n];
    }
    for (int j = 0; k < nslab; ++k, ++j) p[j] += src[(long)k * 48 * n];
    s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
  }
  us[xi][e] = s;
  __syncthreads();
  if (xi != 0 || i >= n) return;
  float u[4][4];
#pragma unroll
  for (int t = 0; t < 16; ++t) u[t >> 2][t & 3] = us[t][e];
  float p[4][3], q[3][3];
#pragma unroll
  for (int y = 0; y < 4; ++y) gt3w(u[y][0], u[y][1], u[y][2], u[y][3], p[y]);
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    float col[3];
    gt3w(p[0][kx], p[1][kx], p[2][kx], p[3][kx], col);
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) q[ky][kx] = col[ky];
  }
  float* dst = dw + i * 27 + a * 9;
#pragma unroll
  for (int t = 0; t < 9; ++t) dst[t] = q[t / 3][t % 3];
}

bool make_plan_w2d(const DramConvDesc* d, WW2Geom& g) {
  if (!d || d->k != 3 || d->stride != 1 || d->dil != 1 || d->pad != 1) return false;
  if (d->B < 1 || d->D < 1 || d->H < 1 || d->W < 1) return false;
  if (d->Do != d->D || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cin % 32 != 0 || d->Cout % 32 != 0 || d->Cin < 32 || d->Cout < 32) return false;
  g.B = d->B; g.D = d->D; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Cout = d->Cout;
  g.ny = (d->H + 7) / 8; g.nx = (d->W + 7) / 8;
  g.ncols = d->B * g.ny * g.nx;
  g.ci_tiles = d->Cin / 32;
  const int pairs = g.ci_tiles * (d->Cout / 32);
  // columns per workgroup: one workgroup per CU (256 slots); maximise useful work / (rounds * 256 * cpw)
  // under a 256 MB slab cap; ties go to fewer slabs
  const double slab1 = 48.0 * d->Cout * d->Cin * 4.0;
  double best = -1.0;
  int best_cpw = g.ncols;
  for (int cpw = g.ncols; cpw >= 1; --cpw) {
    const int ns = (g.ncols + cpw - 1) / cpw;
    if (slab1 * ns > 256e6) break;
    const long wgs = (long)pairs * ns;
    const long rounds = (wgs + 255) / 256;
    const double eff = (double)g.ncols * pairs / ((double)rounds * 256.0 * cpw);
    if (eff > best + 1e-9) { best = eff; best_cpw = cpw; }
  }
  g.cpw = best_cpw;
  g.nslab = (g.ncols + g.cpw - 1) / g.cpw;
  return true;
}

}  // namespace

extern "C" int dram_wgrad_w2d_applicable(const DramConvDesc* d) {
  WW2Geom g{};
  return make_plan_w2d(d, g) ? 1 : 0;
}

extern "C" size_t dram_wgrad_w2d_workspace(const DramConvDesc* d) {
  WW2Geom g{};
  if (!make_plan_w2d(d, g)) return 0;
  return (size_t)g.nslab * 48 * d->Cout * d->Cin * sizeof(float);
}

extern "C" int dram_wgrad_w2d(const float* x, const float* dy, float* dw, const DramConvDesc* d, void* workspace,
                              size_t workspace_bytes, dram_stream_t stream) {
  if (!x || !dy || !dw || !d) return DRAM_ERR_BAD_ARG;
  WW2Geom g{};
  if (!make_plan_w2d(d, g)) return DRAM_ERR_UNSUPPORTED;
  const size_t need = (size_t)g.nslab * 48 * d->Cout * d->Cin * sizeof(float);
  if (!workspace || workspace_bytes < need) return DRAM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int pairs = g.ci_tiles * (d->Cout / 32);
  // kernel + slab reduce in one record; executed: 48 products per 2x2 in-plane tile and channel pair = 12 / voxel
  const double vox = (double)d->B * d->D * d->H * d->W;
  DramProf prof(DRAM_FAM_WGRAD_W2D, 0, 2.0 * vox * d->Cout * d->Cin * 12.0,
                4.0 * (vox * (d->Cin + d->Cout) + 27.0 * d->Cout * d->Cin), s, 2.0 * vox * d->Cout * d->Cin * 27.0);
  hipLaunchKernelGGL(conv_wgrad_w2d_kernel, dim3(pairs * g.nslab), dim3(512), 0, s, x, dy, (float*)workspace, g);
  DRAM_LAUNCH_CHECK();
  const long n = (long)d->Cout * d->Cin;
  hipLaunchKernelGGL(wgrad_w2d_reduce_kernel, dim3((unsigned)((n + 63) / 64), 3), dim3(1024), 0, s,
                     (const float*)workspace, dw, g.nslab, d->Cout, d->Cin);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
