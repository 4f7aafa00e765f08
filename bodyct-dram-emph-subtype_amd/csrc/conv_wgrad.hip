// conv_wgrad.hip -- Conv3d weight gradient on the gfx950 fp32 matrix cores.
//
// Replaces autograd's convolution_backward (weight) for the nn.Conv3d call sites of
// reference med3d.py:91-100, :152-157, :67/:76, :226.
//
//   dW[tap][co][ci] = sum_m dy[m][co] * x[src(m,tap)][ci]      (m = output voxel)
//
// GEMM view per tap: M = co, N = ci, K = voxels.  One workgroup owns a (co-tile, ci-tile,
// kz) triple and a contiguous range of "row chunks" (32 consecutive output x at fixed
// b,z,y); the 3x3 in-plane taps of that kz share one LDS x-tile (3 input rows x
// (31*stride + 2*dil + 1) voxels x 64 ci) and one dy-tile (32 voxels x co-tile), so a
// wave does 9 MFMAs per (1 + 9) LDS dword reads.  Each wave keeps 9 accumulators of
// 32x32 (144 VGPRs).  Operands are read "transposed" straight from the [voxel][channel]
// LDS rows: A[i=co][k=vox] = dy_lds[vox][co] (lanes = consecutive channels: conflict-free).
// Split-K over row chunks writes fp32 slabs; a second kernel sums the slabs in a fixed
// order and scatters into the reference layout [Cout][Cin][kD][kH][kW] -> deterministic.
#include <stdlib.h>
#include "common.h"

namespace {

struct WGeom {
  int B, D, H, W, Cin;
  int Do, Ho, Wo, Cout;
  int k, pad;
  int XC;          // chunks per output row = ceil(Wo/32)
  int NC;          // total chunks = B*Do*Ho*XC
  int co_tiles, ci_tiles;
  int nsplit;
  int cps;         // chunks per split
};

// S: stride, DIL: dilation, K3: 1 -> 3x3x3 (9 in-plane taps per block), 0 -> 1x1x1
// COW x CIW x KW waves (8 = 512 threads, two waves per SIMD so that sibling waves hide each
// other's ds_read latency and barrier skew): co-tile = 32*COW, ci-tile = 32*CIW; the 32 voxels
// of a chunk are split KW ways between wave groups, whose accumulators are folded together
// through LDS (fixed order) once, after the chunk loop.
template <int S, int DIL, int K3, int COW, int CIW, int KW>
__global__ __launch_bounds__(64 * COW * CIW * KW, (COW * CIW * KW) / 4) void conv_wgrad_kernel(
    const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ slab, const WGeom g) {
  constexpr int NW = COW * CIW * KW;
  constexpr int NTHR = 64 * NW;
  static_assert(NW == 8, "8 waves (4-wave workgroups measured 5 % slower: 111.8 vs 117.2 TFLOP/s)");
  constexpr int NT = K3 ? 9 : 1;
  constexpr int NR = K3 ? 3 : 1;
  constexpr int XW = 31 * S + (K3 ? 2 * DIL : 0) + 1;
  constexpr int CO_T = 32 * COW, CI_T = 32 * CIW;
  constexpr int LDY = CO_T + 4, LDX = CI_T + 4;
  constexpr int DYP = (32 * (CO_T / 4) + NTHR - 1) / NTHR;     // dy float4 per thread
  constexpr int XQ = CI_T / 4;                                 // float4 per x voxel row
  constexpr int XP = (NR * XW * XQ + NTHR - 1) / NTHR;         // x float4 per thread
  constexpr int PAIRS = COW * CIW;
  constexpr int LDS_TILE = 32 * LDY + NR * XW * LDX;
  constexpr int LDS_RED = (KW > 1) ? (KW - 1) * PAIRS * 1024 : 0;   // one tap of every folded group
  constexpr int LDSF = LDS_TILE > LDS_RED ? LDS_TILE : LDS_RED;
  __shared__ __attribute__((aligned(16))) float lds[LDSF];
  float* dyl = lds;
  float* xl = lds + 32 * LDY;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cw = wave % COW;
  const int iw = (wave / COW) % CIW;
  const int kwv = wave / (COW * CIW);

  // ---- block decode: (tile, kz, split) --------------------------------------------
  int bid = blockIdx.x;
  const int split = bid % g.nsplit; bid /= g.nsplit;
  const int ci_t = bid % g.ci_tiles; bid /= g.ci_tiles;
  const int co_t = bid % g.co_tiles; bid /= g.co_tiles;
  const int tz = bid;  // 0..k-1
  const int co0 = co_t * CO_T, ci0 = ci_t * CI_T;

  const int q0 = split * g.cps;
  const int q1 = (q0 + g.cps < g.NC) ? q0 + g.cps : g.NC;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

  float4 rdy[DYP];
  float4 rx[XP];

  auto load_chunk = [&](int q) -> bool {
    const int xc = q % g.XC;
    int r = q / g.XC;
    const int yo = r % g.Ho; r /= g.Ho;
    const int zo = r % g.Do;
    const int b = r / g.Do;
    const int zi = zo * S - g.pad + tz * DIL;
    if (zi < 0 || zi >= g.D) return false;  // uniform
    const int xo0 = xc * 32;
    // dy tile: 32 voxels x CO_T
#pragma unroll
    for (int p = 0; p < DYP; ++p) {
      const int idx = p * NTHR + tid;
      const int v = idx / (CO_T / 4), c4 = idx % (CO_T / 4);
      const bool ok = (idx < 32 * (CO_T / 4)) & (xo0 + v < g.Wo);
      const long o = ((((long)b * g.Do + zo) * g.Ho + yo) * g.Wo + xo0 + v) * g.Cout + co0 + c4 * 4;
      rdy[p] = ok ? *reinterpret_cast<const float4*>(dy + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // x tile: NR rows x XW voxels x CI_T
    const int xi0 = xo0 * S - g.pad;
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int idx = p * NTHR + tid;
      const int v = idx / XQ, c4 = idx % XQ;
      const int ty = v / XW, xp = v - ty * XW;
      const int yi = yo * S - g.pad + ty * DIL;
      const int xi = xi0 + xp;
      const bool ok = (idx < NR * XW * XQ) & (yi >= 0) & (yi < g.H) & (xi >= 0) & (xi < g.W);
      const long o = ((((long)b * g.D + zi) * g.H + yi) * g.W + xi) * g.Cin + ci0 + c4 * 4;
      rx[p] = ok ? *reinterpret_cast<const float4*>(x + o) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    return true;
  };

  auto store_chunk = [&]() {
#pragma unroll
    for (int p = 0; p < DYP; ++p) {
      const int idx = p * NTHR + tid;
      const int v = idx / (CO_T / 4), c4 = idx % (CO_T / 4);
      if (idx < 32 * (CO_T / 4)) *reinterpret_cast<float4*>(&dyl[v * LDY + c4 * 4]) = rdy[p];
    }
#pragma unroll
    for (int p = 0; p < XP; ++p) {
      const int idx = p * NTHR + tid;
      const int v = idx / XQ, c4 = idx % XQ;
      if (idx < NR * XW * XQ) *reinterpret_cast<float4*>(&xl[v * LDX + c4 * 4]) = rx[p];
    }
  };

  bool cur = (q0 < q1) ? load_chunk(q0) : false;
  for (int q = q0; q < q1; ++q) {
    __syncthreads();
    if (cur) store_chunk();
    __syncthreads();
    bool nxt = false;
    if (q + 1 < q1) nxt = load_chunk(q + 1);
    if (cur) {
      constexpr int VPW = 32 / KW;  // voxels per wave
#pragma unroll 4
      for (int kk = 0; kk < VPW / 2; ++kk) {
        const int vox = kwv * VPW + 2 * kk + lh;
        const float a = dyl[vox * LDY + cw * 32 + li];
        const float* xb = &xl[(vox * S) * LDX + iw * 32 + li];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
          const int ty = t / 3, tx = t % 3;
          const float bvv = xb[(ty * XW + tx * DIL) * LDX];
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc[t], 0, 0, 0);
        }
      }
    }
    cur = nxt;
  }

  // ---- fold the KW voxel-groups into group 0 (one tap per pass through LDS) -----------
  if (KW > 1) {
    const int pair = cw + COW * iw;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      __syncthreads();
      if (kwv > 0) {
        float* dst = &lds[((kwv - 1) * PAIRS + pair) * 1024 + lane];
#pragma unroll
        for (int e = 0; e < 16; ++e) dst[e * 64] = acc[t][e];
      }
      __syncthreads();
      if (kwv == 0) {
#pragma unroll
        for (int gq = 1; gq < KW; ++gq) {
          const float* src = &lds[((gq - 1) * PAIRS + pair) * 1024 + lane];
#pragma unroll
          for (int e = 0; e < 16; ++e) acc[t][e] += src[e * 64];
        }
      }
    }
    if (kwv != 0) return;
  }

  // ---- write the slab slice ---------------------------------------------------------
  const long tap_stride = (long)g.Cout * g.Cin;
  float* sl = slab + ((long)split * g.k + tz) * NT * tap_stride;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int co = co0 + cw * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int ci = ci0 + iw * 32 + li;
      if (co < g.Cout && ci < g.Cin) sl[t * tap_stride + (long)co * g.Cin + ci] = acc[t][e];
    }
  }
}

// =====================================================================================
// v2 (stride 1, dilation 1): z-walking weight gradient.  A workgroup of 8 waves owns a
// (32 co x 32 ci) block and walks a column of 8x8 output voxels through all Do planes.  LDS
// holds a ring of 4 input planes (10x10 halo rows x 32 ci) and 2 dy planes (64 voxels x 32 co),
// filled by LDS-DMA one plane ahead: one new input plane + one dy plane per step
// (~0.024 vector-memory instructions per MFMA, one barrier per step, no VGPR staging).
// Wave w accumulates taps {w, w+8, w+16, w+24}; operands are ds_read_b32 rows
// (lanes = consecutive channels: conflict-free, no swizzle needed).
__device__ __attribute__((aligned(128))) float g_wzero_line[32];

struct W2Geom {
  int B, D, H, W, Cin, Cout;       // stride 1 / pad 1: output dims == input dims
  int ny, nx, ncols;               // 8x8 columns per plane, total columns = B*ny*nx
  int ci_tiles, nslab, cpw;        // 32-wide ci tiles, workgroups per (co,ci) pair, columns per workgroup
};

__global__ __launch_bounds__(512, 4) void conv_wgrad2_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ dy,
                                                             float* __restrict__ slab, const W2Geom g) {
  constexpr int PR = 104;                 // rows per halo plane slot (100 used, 13 DMA pieces of 8)
  constexpr int HPL = PR * 32;            // floats per halo plane
  constexpr int DPL = 64 * 32;            // floats per dy plane
  __shared__ __attribute__((aligned(1024))) float lds[4 * HPL + 2 * DPL];
  float* halo = lds;
  float* dyl = lds + 4 * HPL;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int sub = lane >> 3, pslot = lane & 7;

  const int pair = blockIdx.x / g.nslab, sidx = blockIdx.x - pair * g.nslab;
  const int co0 = (pair / g.ci_tiles) * 32, ci0 = (pair % g.ci_tiles) * 32;
  const int c_begin = sidx * g.cpw;
  const int c_end = (c_begin + g.cpw < g.ncols) ? c_begin + g.cpw : g.ncols;

  // this wave's taps
  int tbase[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = wave + 8 * j;
    const int ty = (t / 3) % 3, tx = t % 3;
    tbase[j] = (ty * 10 + tx) * 32 + li + lh * 32;   // in-plane operand offset (+ the lh voxel)
  }
  f32x16 acc[4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;

  const float* zline = g_wzero_line + pslot * 4;
  const long plane_x = (long)g.H * g.W * g.Cin, plane_dy = (long)g.H * g.W * g.Cout;

  for (int col = c_begin; col < c_end; ++col) {
    int r = col;
    const int txi = r % g.nx; r /= g.nx;
    const int tyi = r % g.ny;
    const int b = r / g.ny;
    const int y0 = tyi * 8, x0 = txi * 8;
    // in-plane source offsets of this lane's DMA rows (-1: outside the volume -> zero line)
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
      const int yo = y0 + (vox >> 3), xo = x0 + (vox & 7);
      dsrc = (yo < g.H && xo < g.W) ? ((long)yo * g.W + xo) * g.Cout + co0 + pslot * 4 : -1;
    }
    const float* xb = x + (long)b * g.D * plane_x;
    const float* db = dy + (long)b * g.D * plane_dy;

    auto issue_halo = [&](int zi) __attribute__((always_inline)) {   // input plane zi -> ring slot (zi+1)&3
      float* dst = halo + ((zi + 1) & 3) * HPL;
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
    auto issue_dy = [&](int zo) __attribute__((always_inline)) {     // dy plane zo -> slot zo&1
      const float* src = (zo < g.D && dsrc >= 0) ? db + (long)zo * plane_dy + dsrc : zline;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(dyl + (zo & 1) * DPL + wave * 256), 16, 0, 0);
    };

    __syncthreads();                    // previous column fully consumed
    issue_halo(-1); issue_halo(0); issue_halo(1);
    issue_dy(0);
    for (int zo = 0; zo < g.D; ++zo) {
      __syncthreads();                  // planes for step zo have landed; step zo-1 is finished everywhere
      if (zo + 2 <= g.D) issue_halo(zo + 2);   // slot (zo+3)&3 held plane zo-2 (plane D = zero padding)
      if (zo + 1 < g.D) issue_dy(zo + 1);      // slot (zo+1)&1 held dy plane zo-1
      const float* dp = dyl + (zo & 1) * DPL + li + lh * 32;
      const float* hp[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = wave + 8 * j;
        hp[j] = halo + ((zo + t / 9) & 3) * HPL + tbase[j];
      }
#pragma unroll 8
      for (int kk = 0; kk < 32; ++kk) {
        // voxel 2kk+lh of the plane: (y, x) = (kk>>2, 2(kk&3)+lh); the lh part lives in tbase/dp
        const int voff = ((kk >> 2) * 10 + 2 * (kk & 3)) * 32;
        const float a = dp[kk * 64];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if (wave + 8 * j < 27) {      // wave-uniform
            const float bv = hp[j][voff];
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, acc[j], 0, 0, 0);
          }
        }
      }
    }
  }

  // ---- write this workgroup's slab slice: slab[sidx][tap][co][ci] ---------------------------
  const long tap_stride = (long)g.Cout * g.Cin;
  float* sl = slab + (long)sidx * 27 * tap_stride;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int t = wave + 8 * j;
    if (t < 27) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int co = co0 + (e & 3) + 8 * (e >> 2) + 4 * lh;
        sl[t * tap_stride + (long)co * g.Cin + ci0 + li] = acc[j][e];
      }
    }
  }
}

// dW[co][ci][tap] = sum_s slab[s][tap][co][ci]   (one thread per output element: small layers
// have few (co,ci) pairs but hundreds of slabs, so parallelism must come from every element)
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int nslab,
                                    int taps, int Cout, int Cin) {
  const long per = (long)taps * Cout * Cin;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per; i += (long)gridDim.x * blockDim.x) {
    float p[8];                        // eight interleaved partial sums: eight loads in flight, fixed combination order
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = 0.f;
    int k = 0;
    for (; k + 8 <= nslab; k += 8) {
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] += slab[(long)(k + j) * per + i];
    }
    for (int j = 0; k < nslab; ++k, ++j) p[j] += slab[(long)k * per + i];
    const float s = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    const int ci = (int)(i % Cin);
    const long r = i / Cin;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout);
    dw[((long)co * Cin + ci) * taps + tap] = s;
  }
}

struct Plan {
  WGeom g;
  int cow, kw;
  int nslab;
  int nblk;
  int variant;  // 0: s1d1, 1: s1d2, 2: s1d4, 3: s2d1, 4: 1x1x1 (s1)
};

bool make_plan(const DramConvDesc* d, Plan& p) {
  if (!d) return false;
  if (d->k != 1 && d->k != 3) return false;
  if (d->Cin % 64 != 0) return false;
  if (d->Cout % 32 != 0) return false;
  WGeom& g = p.g;
  g.B = d->B; g.D = d->D; g.H = d->H; g.W = d->W; g.Cin = d->Cin;
  g.Do = d->Do; g.Ho = d->Ho; g.Wo = d->Wo; g.Cout = d->Cout;
  g.k = d->k; g.pad = d->pad;
  if (d->k == 3) {
    if (d->stride == 1 && d->dil == 1) p.variant = 0;
    else if (d->stride == 1 && d->dil == 2) p.variant = 1;
    else if (d->stride == 1 && d->dil == 4) p.variant = 2;
    else if (d->stride == 2 && d->dil == 1) p.variant = 3;
    else return false;
  } else {
    if (d->stride != 1) return false;
    p.variant = 4;
  }
  const bool narrow = (d->Cout % 64 != 0);
  p.cow = narrow ? 1 : 2;
  p.kw = 1;   // voxel groups are folded in-kernel: one slab slice per split
  g.co_tiles = d->Cout / (32 * p.cow);
  g.ci_tiles = d->Cin / 64;
  g.XC = (d->Wo + 31) / 32;
  g.NC = d->B * d->Do * d->Ho * g.XC;
  const int tiles = g.co_tiles * g.ci_tiles * d->k;
  // split-K factor: 512..1280 workgroups, chosen so that the grid is as close as possible
  // to a whole number of 256-CU rounds (1026 workgroups would cost a fifth round for 2)
  int ns = 1;
  double best = -1.0;
  for (int c = 1; c <= g.NC && (long)tiles * c <= 1280; ++c) {
    const long wg = (long)tiles * c;
    if (wg < 512 && (long)tiles * (c + 1) <= 1280 && c < g.NC) continue;
    const double eff = (double)wg / (double)(((wg + 255) / 256) * 256);
    if (eff >= best - 1e-9) { best = eff; ns = c; }
  }
  // keep the slab below ~192 MB
  const double slab1 = (double)d->k * d->k * d->k * d->Cout * d->Cin * 4.0 * p.kw;
  while (ns > 1 && slab1 * ns > 192e6) --ns;
  g.cps = (g.NC + ns - 1) / ns;
  ns = (g.NC + g.cps - 1) / g.cps;
  g.nsplit = ns;
  p.nslab = ns * p.kw;
  p.nblk = tiles * ns;
  return true;
}

}  // namespace

// z-walking plan (v2): stride 1, dilation 1, 3x3x3, pad 1, channels multiple of 32
bool make_plan2(const DramConvDesc* d, W2Geom& g) {
  if (const char* e = tune_env("DRAM_WGRAD_V")) { if (e[0] == '1') return false; }
  if (!d || d->k != 3 || d->stride != 1 || d->dil != 1 || d->pad != 1) return false;
  if (d->Cin % 32 != 0 || d->Cout % 32 != 0) return false;
  g.B = d->B; g.D = d->D; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Cout = d->Cout;
  g.ny = (d->H + 7) / 8; g.nx = (d->W + 7) / 8;
  g.ncols = d->B * g.ny * g.nx;
  g.ci_tiles = d->Cin / 32;
  const int pairs = g.ci_tiles * (d->Cout / 32);
  // Columns per workgroup (cpw): every slab is a full [27][Cout][Cin] image, workgroups =
  // pairs * ceil(ncols / cpw) run two per CU (512 slots).  Pick the cpw that maximises
  // useful work / (rounds * 512 * cpw) under a 256 MB slab cap; ties go to fewer slabs.
  const double slab1 = 27.0 * d->Cout * d->Cin * 4.0;
  double best = -1.0;
  int best_cpw = g.ncols;
  for (int cpw = g.ncols; cpw >= 1; --cpw) {
    const int ns = (g.ncols + cpw - 1) / cpw;
    if (slab1 * ns > 256e6) break;
    const long wgs = (long)pairs * ns;
    const long rounds = (wgs + 511) / 512;
    const double eff = (double)g.ncols * pairs / ((double)rounds * 512.0 * cpw);
    if (eff > best + 1e-9) { best = eff; best_cpw = cpw; }
  }
  g.cpw = best_cpw;
  g.nslab = (g.ncols + g.cpw - 1) / g.cpw;
  return true;
}

extern "C" size_t dram_conv3d_bwd_weight_workspace(const DramConvDesc* d) {
  W2Geom g2{};
  if (make_plan2(d, g2)) return (size_t)g2.nslab * 27 * d->Cout * d->Cin * sizeof(float);
  Plan p{};
  if (!make_plan(d, p)) return 0;
  return (size_t)p.nslab * d->k * d->k * d->k * d->Cout * d->Cin * sizeof(float);
}

extern "C" int dram_conv3d_bwd_weight(const float* x, const float* dy, float* dw, const DramConvDesc* d,
                                      void* workspace, size_t workspace_bytes, dram_stream_t stream) {
  Plan p{};
  if (!x || !dy || !dw || !d) return DRAM_ERR_BAD_ARG;
  {
    W2Geom g2{};
    if (make_plan2(d, g2)) {
      const size_t need2 = (size_t)g2.nslab * 27 * d->Cout * d->Cin * sizeof(float);
      if (!workspace || workspace_bytes < need2) return DRAM_ERR_WORKSPACE;
      hipStream_t s2 = (hipStream_t)stream;
      const int pairs = g2.ci_tiles * (d->Cout / 32);
      // one timeline record for kernel + slab reduce: 2*M*N*K executed; x, dy read once, dw written once
      DramProf prof(DRAM_FAM_CONV_WGRAD, 2, 2.0 * d->B * d->Do * d->Ho * d->Wo * (double)d->Cout * d->Cin * 27.0,
                    4.0 * ((double)d->B * d->D * d->H * d->W * d->Cin + (double)d->B * d->Do * d->Ho * d->Wo * d->Cout +
                           27.0 * d->Cout * d->Cin), s2);
      hipLaunchKernelGGL(conv_wgrad2_kernel, dim3(pairs * g2.nslab), dim3(512), 0, s2, x, dy, (float*)workspace, g2);
      DRAM_LAUNCH_CHECK();
      const long per2 = (long)27 * d->Cout * d->Cin;
      const int rg2 = (int)((per2 + 255) / 256 > 8192 ? 8192 : (per2 + 255) / 256);
      hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rg2), dim3(256), 0, s2, (const float*)workspace, dw, g2.nslab, 27,
                         d->Cout, d->Cin);
      DRAM_LAUNCH_CHECK();
      return DRAM_OK;
    }
  }
  if (!make_plan(d, p)) return DRAM_ERR_UNSUPPORTED;
  const size_t need = (size_t)p.nslab * d->k * d->k * d->k * d->Cout * d->Cin * sizeof(float);
  if (!workspace || workspace_bytes < need) return DRAM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* slab = (float*)workspace;
  dim3 grid(p.nblk);
  const bool narrow = (p.cow == 1);
  DramProf prof(DRAM_FAM_CONV_WGRAD, 10 + p.variant,
                2.0 * d->B * d->Do * d->Ho * d->Wo * (double)d->Cout * d->Cin * d->k * d->k * d->k,
                4.0 * ((double)d->B * d->D * d->H * d->W * d->Cin + (double)d->B * d->Do * d->Ho * d->Wo * d->Cout +
                       (double)d->k * d->k * d->k * d->Cout * d->Cin), s);
#define WG_LAUNCH(S_, D_, K3_)                                                                              \
  do {                                                                                                      \
    if (narrow)                                                                                             \
      hipLaunchKernelGGL((conv_wgrad_kernel<S_, D_, K3_, 1, 2, 4>), grid, dim3(512), 0, s, x, dy, slab, p.g); \
    else                                                                                                    \
      hipLaunchKernelGGL((conv_wgrad_kernel<S_, D_, K3_, 2, 2, 2>), grid, dim3(512), 0, s, x, dy, slab, p.g); \
  } while (0)
  switch (p.variant) {
    case 0: WG_LAUNCH(1, 1, 1); break;
    case 1: WG_LAUNCH(1, 2, 1); break;
    case 2: WG_LAUNCH(1, 4, 1); break;
    case 3: WG_LAUNCH(2, 1, 1); break;
    case 4: WG_LAUNCH(1, 1, 0); break;
    default: return DRAM_ERR_UNSUPPORTED;
  }
#undef WG_LAUNCH
  DRAM_LAUNCH_CHECK();
  const int taps = d->k * d->k * d->k;
  const long per = (long)taps * d->Cout * d->Cin;
  const int rgrid = (int)((per + 255) / 256 > 8192 ? 8192 : (per + 255) / 256);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(rgrid), dim3(256), 0, s, slab, dw, p.nslab, taps, d->Cout, d->Cin);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
