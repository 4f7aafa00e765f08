// conv_wino.hip -- Winograd F(2x2x2, 3x3x3) path for the wide stride-1 3x3x3 convolutions
// (forward, data gradient and weight gradient) on the gfx950 fp32 matrix cores.
//
// Same call sites as conv_igemm.hip / conv_wgrad.hip (reference med3d.py:91-100 conv3x3x3 inside
// BasicBlock / Bottleneck, dilation 1/2/4 with padding == dilation; autograd's
// convolution_backward for them).  The direct implicit GEMM already runs at ~0.86 of the fp32
// MFMA peak, so the remaining lever is arithmetic: the 3-D Winograd transform needs 64 instead
// of 216 multiplies per 2x2x2 output tile and channel pair (3.375x fewer MFMA flops).
//
//   U[xi][co][ci] = G w G^T (3-D)                      weight transform, once per step
//   V[xi][t][ci]  = B^T x B (3-D) of the 4x4x4 input tile t            wino_in_kernel<0>
//   M[xi][t][co]  = sum_ci V[xi][t][ci] * U[xi][co][ci]   64 dense GEMMs  wino_gemm_nn_kernel
//   y(tile t)     = A^T M A (3-D) + bias (+ fused epilogue, BN partial sums) wino_out_kernel
// Data gradient: the same pipeline on dy with the tap-flipped, transposed weights.
// Weight gradient: dU[xi][co][ci] = sum_t (A dy A^T)[xi][t][co] * V[xi][t][ci]  (TN GEMMs, split
// over t into slabs), then dw = G^T (sum of slabs) G in a fixed order -> deterministic.
//
// A dilated convolution is d^3 independent dilation-1 convolutions on the residue sub-lattices,
// so the tiles are 2x2x2 blocks of a sub-lattice (voxel step d).  V and M round-trip through HBM
// (8x the activation bytes), which is why the path is planned only for >= 256-channel layers:
// there the GEMMs dominate (layer4: 2.3 GB of transform traffic vs 137 GFLOP per pass).
//
// GEMM kernels: 512 threads (8 waves as 4(M) x 2(N)), tile 256 x (64*NJ) x 32, operands staged by
// LDS-DMA (global_load_lds_dwordx4), double-buffered, one barrier per K-step; NN form reads both
// operands with the ds_read_b128 k-permutation trick of conv_igemm.hip, TN form reads [t][c] rows
// with ds_read_b32 (lanes = consecutive channels).
#include <stdlib.h>
#include "common.h"

namespace {

struct WinoGeom {
  int B, D, H, W;  // voxel grid (input and output grids coincide: stride 1, pad == dil)
  int d;           // dilation
  int Tz, Ty, Tx;  // 2x2x2 tiles per residue sub-lattice axis
  int T;           // B * d^3 * Tz * Ty * Tx
  int Tpad;        // T rounded up to the GEMM M tile (256)
};

// first OUTPUT voxel of tile t; tile-local positions step by g.d
__device__ __forceinline__ void tile_origin(const WinoGeom& g, int t, int& b, int& z0, int& y0, int& x0) {
  int r = t;
  const int tx = r % g.Tx; r /= g.Tx;
  const int ty = r % g.Ty; r /= g.Ty;
  const int tz = r % g.Tz; r /= g.Tz;
  const int rx = r % g.d; r /= g.d;
  const int ry = r % g.d; r /= g.d;
  const int rz = r % g.d;
  b = r / g.d;
  z0 = 2 * tz * g.d + rz;
  y0 = 2 * ty * g.d + ry;
  x0 = 2 * tx * g.d + rx;
}

// ------------------------------------------------------------------------------------------
// Tile transforms into the Winograd domain.  One wave per (tile, 64-channel block); lanes are
// consecutive channels (256-B coalesced rows).  out[xi][t][c], xi = (i*4 + j)*4 + k.
//   MODE 0:  V = B^T v B over the 4x4x4 input tile (zero outside the volume)
//   MODE 1:  A dy A^T over the 2x2x2 output-gradient tile (weight gradient)
// Rows t in [T, Tpad) are written as zeros (the TN GEMM contracts over t).
template <int MODE>
__global__ __launch_bounds__(256) void wino_in_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                      const WinoGeom g, const int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cblks = C >> 6;
  const long plane = (long)g.Tpad * C;
  const long total = (long)g.Tpad * cblks;
  for (long w = blockIdx.x * 4L + wave; w < total; w += gridDim.x * 4L) {
    const int t = (int)(w / cblks);
    const int c = (int)(w - (long)t * cblks) * 64 + lane;
    float v[4][4][4];
    if (t >= g.T) {
#pragma unroll
      for (int i = 0; i < 64; ++i) out[i * plane + (long)t * C + c] = 0.f;
      continue;
    }
    int b, z0, y0, x0;
    tile_origin(g, t, b, z0, y0, x0);
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int z = z0 + (i - 1) * g.d;
        const bool zo = (z >= 0) & (z < g.D);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int y = y0 + (j - 1) * g.d;
          const bool yo = zo & (y >= 0) & (y < g.H);
#pragma unroll
          for (int k = 0; k < 4; ++k) {
            const int x = x0 + (k - 1) * g.d;
            const bool ok = yo & (x >= 0) & (x < g.W);
            const long o = ((((long)b * g.D + z) * g.H + y) * g.W + x) * C + c;
            v[i][j][k] = ok ? in[o] : 0.f;
          }
        }
      }
      // B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1] along x, y, z
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float a0 = v[i][j][0], a1 = v[i][j][1], a2 = v[i][j][2], a3 = v[i][j][3];
          v[i][j][0] = a0 - a2; v[i][j][1] = a1 + a2; v[i][j][2] = a2 - a1; v[i][j][3] = a1 - a3;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a0 = v[i][0][k], a1 = v[i][1][k], a2 = v[i][2][k], a3 = v[i][3][k];
          v[i][0][k] = a0 - a2; v[i][1][k] = a1 + a2; v[i][2][k] = a2 - a1; v[i][3][k] = a1 - a3;
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a0 = v[0][j][k], a1 = v[1][j][k], a2 = v[2][j][k], a3 = v[3][j][k];
          v[0][j][k] = a0 - a2; v[1][j][k] = a1 + a2; v[2][j][k] = a2 - a1; v[3][j][k] = a1 - a3;
        }
    } else {
      float u[2][2][2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int z = z0 + i * g.d;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int y = y0 + j * g.d;
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int x = x0 + k * g.d;
            const bool ok = (z < g.D) & (y < g.H) & (x < g.W);
            const long o = ((((long)b * g.D + z) * g.H + y) * g.W + x) * C + c;
            u[i][j][k] = ok ? in[o] : 0.f;
          }
        }
      }
      // A = [1 0; 1 1; 1 -1; 0 -1] along x, y, z
      float p[2][2][4], q[2][4][4];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float a0 = u[i][j][0], a1 = u[i][j][1];
          p[i][j][0] = a0; p[i][j][1] = a0 + a1; p[i][j][2] = a0 - a1; p[i][j][3] = -a1;
        }
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a0 = p[i][0][k], a1 = p[i][1][k];
          q[i][0][k] = a0; q[i][1][k] = a0 + a1; q[i][2][k] = a0 - a1; q[i][3][k] = -a1;
        }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float a0 = q[0][j][k], a1 = q[1][j][k];
          v[0][j][k] = a0; v[1][j][k] = a0 + a1; v[2][j][k] = a0 - a1; v[3][j][k] = -a1;
        }
    }
    float* o = out + (long)t * C + c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) o[((i * 4 + j) * 4 + k) * plane] = v[i][j][k];
  }
}

// ------------------------------------------------------------------------------------------
// Output transform: y(tile) = A^T M A (3-D), + bias, optional fused  += add * (gate > 0)
// (identity-shortcut gradient), per-channel BatchNorm partial sums.  A workgroup owns
// TPB consecutive tiles x 64 channels; stats row = tile block.
constexpr int WINO_TPB = 16;

__global__ __launch_bounds__(256) void wino_out_kernel(const float* __restrict__ mh, const float* __restrict__ bias,
                                                       const float* __restrict__ add, const float* __restrict__ gate,
                                                       float* __restrict__ out, float* __restrict__ stats,
                                                       const WinoGeom g, const int N) {
  __shared__ float red[4][2][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int cblks = N >> 6;
  const int cb = blockIdx.x % cblks, tb = blockIdx.x / cblks;
  const int c = cb * 64 + lane;
  const long plane = (long)g.Tpad * N;
  const float bv = bias ? bias[c] : 0.f;
  float s1 = 0.f, s2 = 0.f;
  for (int q = wave; q < WINO_TPB; q += 4) {
    const int t = tb * WINO_TPB + q;
    if (t >= g.T) break;
    float m[4][4][4];
    const float* src = mh + (long)t * N + c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) m[i][j][k] = src[((i * 4 + j) * 4 + k) * plane];
    // A^T = [1 1 1 0; 0 1 -1 -1] along x, y, z
    float p[4][4][2], q2[4][2][2], o[2][2][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        p[i][j][0] = m[i][j][0] + m[i][j][1] + m[i][j][2];
        p[i][j][1] = m[i][j][1] - m[i][j][2] - m[i][j][3];
      }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        q2[i][0][k] = p[i][0][k] + p[i][1][k] + p[i][2][k];
        q2[i][1][k] = p[i][1][k] - p[i][2][k] - p[i][3][k];
      }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        o[0][j][k] = q2[0][j][k] + q2[1][j][k] + q2[2][j][k];
        o[1][j][k] = q2[1][j][k] - q2[2][j][k] - q2[3][j][k];
      }
    int b, z0, y0, x0;
    tile_origin(g, t, b, z0, y0, x0);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int z = z0 + i * g.d;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int y = y0 + j * g.d;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
          const int x = x0 + k * g.d;
          if ((z < g.D) & (y < g.H) & (x < g.W)) {
            const long oo = ((((long)b * g.D + z) * g.H + y) * g.W + x) * N + c;
            float v = o[i][j][k] + bv;
            if (add) {
              const float av = add[oo];
              v += gate ? (gate[oo] > 0.f ? av : 0.f) : av;
            }
            out[oo] = v;
            s1 += v;
            s2 += v * v;
          }
        }
      }
    }
  }
  if (stats) {
    red[wave][0][lane] = s1;
    red[wave][1][lane] = s2;
    __syncthreads();
    if (threadIdx.x < 128) {
      const int which = threadIdx.x >> 6;
      const float v = red[0][which][lane] + red[1][which][lane] + red[2][which][lane] + red[3][which][lane];
      stats[((long)tb * 2 + which) * N + c] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Weight transform  U = G w G^T (3-D), G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1].
//   blockIdx.y == 0: uf[xi][co][ci]               (forward B operand, K = ci contiguous)
//   blockIdx.y == 1: ub[xi][ci][co], taps flipped  (data-gradient B operand, K = co contiguous)
__device__ __forceinline__ void g4(const float a0, const float a1, const float a2, float& r0, float& r1, float& r2,
                                   float& r3) {
  const float h = 0.5f * (a0 + a2);
  r0 = a0;
  r1 = h + 0.5f * a1;
  r2 = h - 0.5f * a1;
  r3 = a2;
}

__global__ __launch_bounds__(256) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ uf,
                                                          float* __restrict__ ub, const int Cout, const int Cin) {
  const bool bwd = blockIdx.y == 1;
  float* dst = bwd ? ub : uf;
  if (!dst) return;
  const long n = (long)Cout * Cin;
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= n) return;
  int co, ci;
  if (!bwd) { ci = (int)(i % Cin); co = (int)(i / Cin); }
  else { co = (int)(i % Cout); ci = (int)(i / Cout); }
  const float* src = w + ((long)co * Cin + ci) * 27;
  float gw[3][3][3];
#pragma unroll
  for (int a = 0; a < 27; ++a) (&gw[0][0][0])[a] = src[bwd ? 26 - a : a];
  float p[3][3][4], q[3][4][4], u[4][4][4];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) g4(gw[a][b][0], gw[a][b][1], gw[a][b][2], p[a][b][0], p[a][b][1], p[a][b][2], p[a][b][3]);
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int k = 0; k < 4; ++k) g4(p[a][0][k], p[a][1][k], p[a][2][k], q[a][0][k], q[a][1][k], q[a][2][k], q[a][3][k]);
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int k = 0; k < 4; ++k) g4(q[0][j][k], q[1][j][k], q[2][j][k], u[0][j][k], u[1][j][k], u[2][j][k], u[3][j][k]);
#pragma unroll
  for (int a = 0; a < 64; ++a) dst[a * n + i] = (&u[0][0][0])[a];
}

// dw[co][ci][27] = G^T (sum over splits of slab[split][xi][co][ci]) G   (3-D), fixed order
__device__ __forceinline__ void gt3(const float a0, const float a1, const float a2, const float a3, float& r0,
                                    float& r1, float& r2) {
  r0 = a0 + 0.5f * (a1 + a2);
  r1 = 0.5f * (a1 - a2);
  r2 = 0.5f * (a1 + a2) + a3;
}

// One workgroup = 64 (co, ci) elements x 16 point groups (1024 threads): every thread sums 4 of the
// 64 points over the splits (coalesced over elements), then 64 threads apply G^T . G (3-D).
__global__ __launch_bounds__(1024) void wino_wgrad_out_kernel(const float* __restrict__ slab, float* __restrict__ dw,
                                                               const int Cout, const int Cin, const int nsplit) {
  __shared__ float us[64][64];
  const long n = (long)Cout * Cin;
  const int e = threadIdx.x & 63, gq = threadIdx.x >> 6;
  const long i = blockIdx.x * 64L + e;   // (co, ci), ci fastest
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int a = gq * 4 + j;
    float acc = 0.f;
    if (i < n)
      for (int sp = 0; sp < nsplit; ++sp) acc += slab[((long)sp * 64 + a) * n + i];
    us[a][e] = acc;
  }
  __syncthreads();
  if (gq != 0 || i >= n) return;
  float s[4][4][4];
#pragma unroll
  for (int a = 0; a < 64; ++a) (&s[0][0][0])[a] = us[a][e];
  float p[4][4][3], q[4][3][3], r[3][3][3];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) gt3(s[a][b][0], s[a][b][1], s[a][b][2], s[a][b][3], p[a][b][0], p[a][b][1], p[a][b][2]);
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int k = 0; k < 3; ++k) gt3(p[a][0][k], p[a][1][k], p[a][2][k], p[a][3][k], q[a][0][k], q[a][1][k], q[a][2][k]);
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int k = 0; k < 3; ++k) gt3(q[0][j][k], q[1][j][k], q[2][j][k], q[3][j][k], r[0][j][k], r[1][j][k], r[2][j][k]);
  float* dst = dw + i * 27;
#pragma unroll
  for (int a = 0; a < 27; ++a) dst[a] = (&r[0][0][0])[a];
}

// ------------------------------------------------------------------------------------------
// NN batched GEMM:  Y[xi][m][n] = sum_k A[xi][m][k] * Bw[xi][n][k]     (M = Tpad, K % 32 == 0)
template <int NJ>
__global__ __launch_bounds__(512) void wino_gemm_nn_kernel(const float* __restrict__ A, const float* __restrict__ Bw,
                                                           float* __restrict__ Y, const int Mpad, const int N,
                                                           const int K, const int m_tiles, const int n_tiles,
                                                           const int nblk) {
  constexpr int BN = 64 * NJ;
  constexpr int STAGE = (256 + BN) * 32;
  __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, nblk);
  const int nt = L % n_tiles;
  const int r0 = L / n_tiles;
  const int mt = r0 % m_tiles;
  const int xi = r0 / m_tiles;
  const float* Ab = A + ((long)xi * Mpad + (long)mt * 256) * K;
  const float* Bb = Bw + ((long)xi * N + (long)nt * BN) * K;
  float* Yb = Y + ((long)xi * Mpad + (long)mt * 256) * N + nt * BN;

  // DMA pieces (8 rows x 128 B each): A rows 32*wave + 8j + sub, B rows 8*NJ*wave + 8jj + sub.
  // 16-B slot swizzle slot ^ ((row >> 1) & 7) applied on the source address.
  const int sub = lane >> 3, pslot = lane & 7;
  const int s_even = pslot ^ (lane >> 4), s_odd = s_even ^ 4;
  int aoff[4], boff[NJ];
#pragma unroll
  for (int j = 0; j < 4; ++j) aoff[j] = (32 * wave + 8 * j + sub) * K + ((j & 1) ? s_odd : s_even) * 4;
#pragma unroll
  for (int jj = 0; jj < NJ; ++jj) {
    const int nrow = 8 * NJ * wave + 8 * jj + sub;
    boff[jj] = nrow * K + (pslot ^ ((nrow >> 1) & 7)) * 4;
  }
  auto issue = [&](int it, int stage) __attribute__((always_inline)) {
    float* as = lds + stage * STAGE + 32 * wave * 32;
    float* bs = lds + stage * STAGE + 256 * 32 + 8 * NJ * wave * 32;
    const float* ag = Ab + it * 32;
    const float* bg = Bb + it * 32;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ag + aoff[j]),
                                       (__attribute__((address_space(3))) void*)(as + j * 8 * 32), 16, 0, 0);
#pragma unroll
    for (int jj = 0; jj < NJ; ++jj)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bg + boff[jj]),
                                       (__attribute__((address_space(3))) void*)(bs + jj * 8 * 32), 16, 0, 0);
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave & 3, wn = wave >> 2;
  const int rsw = (li >> 1) & 7;
  const int a_row = (wm * 64 + li) * 32;
  const int b_row = 256 * 32 + (wn * NJ * 32 + li) * 32;
  const int niter = K / 32;

  issue(0, 0);
  for (int it = 0; it < niter; ++it) {
    __syncthreads();   // tile `it` has landed; stage (it+1)&1 is free again
    if (it + 1 < niter) issue(it + 1, (it + 1) & 1);
    const float* st = lds + (it & 1) * STAGE;
#pragma unroll
    for (int gk = 0; gk < 4; ++gk) {
      const int so = ((2 * gk + lh) ^ rsw) * 4;
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(st + a_row + so);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(st + a_row + 32 * 32 + so);
      f32x4 bf[NJ];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) bf[nj] = *reinterpret_cast<const f32x4*>(st + b_row + nj * 32 * 32 + so);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          acc[0][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[e], bf[nj][e], acc[0][nj], 0, 0, 0);
          acc[1][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], bf[nj][e], acc[1][nj], 0, 0, 0);
        }
      }
    }
  }

#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      float* o = Yb + (long)row * N + wn * NJ * 32 + li;
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) o[nj * 32] = acc[mi][nj][e];
    }
}

// ------------------------------------------------------------------------------------------
// TN batched GEMM (weight gradient):
//   slab[split][xi][m][n] = sum_{t in split} Ah[xi][t][m] * Bh[xi][t][n]
// 8 waves as WMW (M) x 8/WMW (N), wave tile 32*MI x 32*NJ; M % (WMW*32*MI) == 0, N may be ragged
// (the DMA column is clamped into the row, the extra columns are never stored).
template <int WMW, int MI, int NJ>
__global__ __launch_bounds__(512) void wino_gemm_tn_kernel(const float* __restrict__ Ah, const float* __restrict__ Bh,
                                                           float* __restrict__ slab, const int Tpad, const int M,
                                                           const int N, const int m_tiles, const int n_tiles,
                                                           const int nsplit, const int kper, const int nblk) {
  constexpr int WNW = 8 / WMW;
  constexpr int BM = WMW * 32 * MI, BN = WNW * 32 * NJ;
  static_assert(BM % 64 == 0 && BM <= 256 && BN % 64 == 0 && BN <= 256, "one DMA piece = 256 floats");
  constexpr int STAGE = 32 * (BM + BN);
  constexpr int AQ = BM / 4, ARPP = 64 / AQ, APW = BM / 64;   // 16-B slots per row, rows per piece, pieces per wave
  constexpr int BQ = BN / 4, BRPP = 64 / BQ, BPW = BN / 64;
  __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int L = xcd_remap(blockIdx.x, nblk);
  const int nt = L % n_tiles; L /= n_tiles;
  const int mt = L % m_tiles; L /= m_tiles;
  const int split = L % nsplit;
  const int xi = L / nsplit;
  const int t0 = split * kper;
  const int t1 = (t0 + kper < Tpad) ? t0 + kper : Tpad;
  int bcol = nt * BN + (lane % BQ) * 4;
  if (bcol > N - 4) bcol = N - 4;
  const float* Ab = Ah + ((long)xi * Tpad + t0 + lane / AQ) * M + mt * BM + (lane % AQ) * 4;
  const float* Bb = Bh + ((long)xi * Tpad + t0 + lane / BQ) * N + bcol;

  auto issue = [&](int it, int stage) __attribute__((always_inline)) {
    float* as = lds + stage * STAGE;
    float* bs = as + 32 * BM;
    const float* ag = Ab + (long)it * 32 * M;
    const float* bg = Bb + (long)it * 32 * N;
#pragma unroll
    for (int j = 0; j < APW; ++j) {
      const int p = APW * wave + j;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(ag + (long)(p * ARPP) * M),
                                       (__attribute__((address_space(3))) void*)(as + p * 256), 16, 0, 0);
    }
#pragma unroll
    for (int jj = 0; jj < BPW; ++jj) {
      const int p = BPW * wave + jj;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(bg + (long)(p * BRPP) * N),
                                       (__attribute__((address_space(3))) void*)(bs + p * 256), 16, 0, 0);
    }
  };

  f32x16 acc[MI][NJ];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const int wm = wave % WMW, wn = wave / WMW;
  const int niter = (t1 - t0) / 32;

  if (niter > 0) issue(0, 0);
  for (int it = 0; it < niter; ++it) {
    __syncthreads();
    if (it + 1 < niter) issue(it + 1, (it + 1) & 1);
    const float* as = lds + (it & 1) * STAGE + wm * 32 * MI + li;
    const float* bs = lds + (it & 1) * STAGE + 32 * BM + wn * 32 * NJ + li;
#pragma unroll 4
    for (int kk = 0; kk < 16; ++kk) {
      const int kr = 2 * kk + lh;
      float af[MI], bf[NJ];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) af[mi] = as[kr * BM + mi * 32];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj) bf[nj] = bs[kr * BN + nj * 32];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          acc[mi][nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[mi], bf[nj], acc[mi][nj], 0, 0, 0);
    }
  }

  float* sb = slab + (((long)split * 64 + xi) * M + mt * BM) * N + nt * BN;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wm * 32 * MI + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      float* o = sb + (long)row * N + wn * 32 * NJ + li;
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
        if (nt * BN + wn * 32 * NJ + nj * 32 + li < N) o[nj * 32] = acc[mi][nj][e];
    }
}

// ------------------------------------------------------------------------------------------
// host side
bool wino_geom_ok(const DramConvDesc* d) {
  if (!d) return false;
  if (d->B < 1 || d->D < 1 || d->H < 1 || d->W < 1) return false;
  if (d->k != 3 || d->stride != 1 || d->dil < 1 || d->pad != d->dil) return false;
  if (d->Do != d->D || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cin < 64 || d->Cout < 64 || d->Cin % 64 != 0 || d->Cout % 64 != 0) return false;
  return true;
}

WinoGeom make_geom(const DramConvDesc* d) {
  WinoGeom g{};
  g.B = d->B; g.D = d->D; g.H = d->H; g.W = d->W; g.d = d->dil;
  auto tiles = [&](int n) { return ((n + g.d - 1) / g.d + 1) / 2; };
  g.Tz = tiles(g.D); g.Ty = tiles(g.H); g.Tx = tiles(g.W);
  const long long T = (long long)g.B * g.d * g.d * g.d * g.Tz * g.Ty * g.Tx;
  g.T = (int)T;
  g.Tpad = (int)((T + 255) / 256 * 256);
  return g;
}

bool wino_size_ok(const DramConvDesc* d) {   // int32 offsets inside one xi plane of the GEMM operands
  const WinoGeom g = make_geom(d);
  const long long T = (long long)g.B * g.d * g.d * g.d * g.Tz * g.Ty * g.Tx;
  const long long cmax = d->Cin > d->Cout ? d->Cin : d->Cout;
  return T > 0 && (T + 255) * cmax < (1LL << 31);
}

int nj_for(int N) { return N % 256 == 0 ? 4 : (N % 128 == 0 ? 2 : 1); }

// wgrad: M = Cout, N = Cin.  Tile = the largest (BM, BN) that divides M and wastes < 13 % of N;
// split over t so that >= ~512 workgroups are in flight.
struct TnPlan { int bm, bn, m_tiles, n_tiles, nsplit, kper; };
bool plan_tn(const DramConvDesc* d, const WinoGeom& g, TnPlan& p) {
  const int M = d->Cout, N = d->Cin;
  p.bm = M % 256 == 0 ? 256 : (M % 128 == 0 ? 128 : 64);
  const int ncand = p.bm == 256 ? 3 : 2;             // 64-column tiles only exist for BM = 256
  const int cand[3] = {256, 128, 64};
  p.bn = 0;
  int best_pad = 1 << 30;
  for (int i = 0; i < ncand; ++i) {
    const int padded = (N + cand[i] - 1) / cand[i] * cand[i];
    if (padded * 100 <= N * 113) { p.bn = cand[i]; break; }
    if (padded < best_pad) { best_pad = padded; p.bn = cand[i]; }
  }
  p.m_tiles = M / p.bm;
  p.n_tiles = (N + p.bn - 1) / p.bn;
  const int base = 64 * p.m_tiles * p.n_tiles;
  const int k32 = g.Tpad / 32;
  int ns = 1;
  while (base * ns < 512 && ns * 2 <= k32 / 4 && ns < 16) ns *= 2;
  p.nsplit = ns;
  p.kper = ((k32 + ns - 1) / ns) * 32;
  return true;
}

int grid_for(long waves) {
  long b = (waves + 3) / 4;
  return (int)(b > 65536 ? 65536 : (b < 1 ? 1 : b));
}

int run_nn(const float* A, const float* U, float* Y, const WinoGeom& g, int N, int K, hipStream_t s) {
  const int nj = nj_for(N);
  const int m_tiles = g.Tpad / 256, n_tiles = N / (64 * nj);
  const int nblk = 64 * m_tiles * n_tiles;
#define WNN(NJ_) \
  hipLaunchKernelGGL((wino_gemm_nn_kernel<NJ_>), dim3(nblk), dim3(512), 0, s, A, U, Y, g.Tpad, N, K, m_tiles, n_tiles, nblk)
  if (nj == 4) WNN(4);
  else if (nj == 2) WNN(2);
  else WNN(1);
#undef WNN
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

// shared by forward (x, uf) and data gradient (dy, ub): in[..., K] -> out[..., N]
int run_conv(const float* in, const float* U, const float* bias, const float* add, const float* gate, float* out,
             float* stats, float* v_keep, const DramConvDesc* d, int K, int N, void* ws, size_t ws_bytes,
             hipStream_t s) {
  const WinoGeom g = make_geom(d);
  const size_t need = (size_t)64 * g.Tpad * ((size_t)K + N) * sizeof(float);
  if (!ws || ws_bytes < need) return DRAM_ERR_WORKSPACE;
  float* V = v_keep ? v_keep : (float*)ws;          // kept for the weight gradient when the caller asks
  float* Mh = (float*)ws + (size_t)64 * g.Tpad * K;
  hipLaunchKernelGGL((wino_in_kernel<0>), dim3(grid_for((long)g.Tpad * (K / 64))), dim3(256), 0, s, in, V, g, K);
  DRAM_LAUNCH_CHECK();
  const int rc = run_nn(V, U, Mh, g, N, K, s);
  if (rc != DRAM_OK) return rc;
  const int ntb = (g.T + WINO_TPB - 1) / WINO_TPB;
  hipLaunchKernelGGL(wino_out_kernel, dim3(ntb * (N / 64)), dim3(256), 0, s, Mh, bias, add, gate, out, stats, g, N);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------
extern "C" int dram_wino_applicable(const DramConvDesc* d) {
  return (wino_geom_ok(d) && wino_size_ok(d)) ? 1 : 0;
}

// Plan: a cost model calibrated on one MI355X (rocprofv3, round 1).  The direct implicit GEMM
// runs at ~135 TFLOP/s; the Winograd pipeline costs, per voxel of the padded tile grid, the
// input transform (36 B x K at ~4.9 TB/s), the 64 GEMMs (16 K N flops at ~118 TFLOP/s, or their
// 32 (K + N) B of operand traffic at ~4.7 TB/s, whichever is larger) and the output transform
// (36 B x N at ~3.4 TB/s).  Forward and data gradient (K and N swapped) are averaged; Winograd is
// planned when it is at least 8 % cheaper.  DRAM_CONV_ALGO: 0/unset auto, 1 always direct,
// 2 Winograd wherever applicable (tests).
// Small volumes underfill the chip: a launch of `wgs` workgroups on `slots` resident slots runs at
// wgs / (rounds * slots) of the steady-state rate.
static double fill(double wgs, double slots) {
  if (wgs < 1.0) wgs = 1.0;
  const double rounds = (double)(long)((wgs + slots - 1.0) / slots);
  return wgs / (rounds * slots);
}
static double wino_cost_per_voxel(double K, double N, double tpad) {
  const double nt = (double)(long)((N + 255.0) / 256.0);           // 256-column GEMM tiles (fewer columns: one tile)
  const double gemm = 16.0 * K * N / (118e12 * fill(64.0 * (tpad / 256.0) * nt, 256.0));
  const double traffic = 32.0 * (K + N) / 4.7e12;
  return 36.0 * K / 4.9e12 + (gemm > traffic ? gemm : traffic) + 36.0 * N / 3.4e12;
}
// direct implicit GEMM: 8x8x8 / 4x8x8 output tiles x 64..256 columns, two workgroups per CU
static double direct_rate(const DramConvDesc* d, double N) {
  const double lat = d->dil;
  const double sz = (double)(long)((d->D + lat - 1) / lat), sy = (double)(long)((d->H + lat - 1) / lat),
               sx = (double)(long)((d->W + lat - 1) / lat);
  const double mt = d->B * lat * lat * lat * (double)(long)((sz + 3) / 4) * (double)(long)((sy + 7) / 8) *
                    (double)(long)((sx + 7) / 8);
  return 135e12 * fill(mt * (double)(long)((N + 127.0) / 128.0), 512.0);
}

extern "C" int dram_conv_algo(const DramConvDesc* d) {
  const char* v = getenv("DRAM_CONV_ALGO");   // read per call: tests switch it between cases
  const int algo = v ? atoi(v) : 0;
  if (algo == 1) return 0;
  const bool w3 = dram_wino_applicable(d) != 0, w2 = dram_wino2d_applicable(d) != 0;
  if (algo == 2) return w3 ? 1 : 0;
  if (algo == 3) return w2 ? 2 : 0;
  const double vox = d ? (double)d->B * d->D * d->H * d->W : 0.0;
  double best = 1e30;
  int pick = 0;
  if (w3) {
    const WinoGeom g = make_geom(d);
    if (g.T >= 128) {
      const double vpad = 8.0 * g.Tpad;
      const double direct = vox * 54.0 * d->Cin * d->Cout * 0.5 *
                            (1.0 / direct_rate(d, d->Cout) + 1.0 / direct_rate(d, d->Cin));
      const double wino = vpad * 0.5 * (wino_cost_per_voxel(d->Cin, d->Cout, g.Tpad) +
                                        wino_cost_per_voxel(d->Cout, d->Cin, g.Tpad));
      if (wino < 0.92 * direct) { best = wino; pick = 1; }
    }
  }
  if (w2 && d->D >= 12 && vox >= 65536.0) {
    // fused in-plane Winograd: measured ~200 TFLOP/s of direct-conv-equivalent work with 64-column
    // tiles, ~155 with 32-column tiles (forward writes Cout columns, data gradient Cin columns)
    const int zt = (d->D + 15) / 16 * 16;
    const double rf = d->Cout % 64 == 0 ? 200e12 : 155e12, rb = d->Cin % 64 == 0 ? 200e12 : 155e12;
    const double w2d = vox * zt / d->D * 54.0 * d->Cin * d->Cout * 0.5 * (1.0 / rf + 1.0 / rb);
    const double direct = vox * 54.0 * d->Cin * d->Cout / 135e12;
    if (w2d < 0.92 * direct && w2d < best) { best = w2d; pick = 2; }
  }
  return pick;
}

constexpr double W2D_WGRAD_RATE = 215e12;   // measured 197-248 TFLOP/s direct-equivalent (round 1)

// Weight-gradient plan (independent of the forward plan: the fused in-plane kernel has no weight
// gradient of its own): 1 = Winograd TN pipeline (dram_wino_conv3d_bwd_weight), 0 = direct.
// Cost per voxel: both tile transforms (36 B x (Cin + Cout) at ~4.9 TB/s) + the 64 TN GEMMs
// (16 Cin Cout flops at ~125 TFLOP/s or their 32 (Cin + Cout) B of operands at ~4.7 TB/s).
extern "C" int dram_conv_wgrad_algo(const DramConvDesc* d) {
  const char* v = getenv("DRAM_CONV_ALGO");
  const int algo = v ? atoi(v) : 0;
  if (algo == 1) return 0;
  const bool w3 = dram_wino_applicable(d) != 0, w2 = dram_wgrad_w2d_applicable(d) != 0;
  if (algo == 2) return w3 ? 1 : 0;
  if (algo == 3) return w2 ? 2 : 0;
  if (!d) return 0;
  const double vox = (double)d->B * d->D * d->H * d->W;
  const double K = d->Cin, N = d->Cout;
  const double direct = vox * 54.0 * K * N / 125e12;
  double best = 1e30;
  int pick = 0;
  if (w3) {
    const WinoGeom g = make_geom(d);
    if (g.T >= 128) {
      const double vpad = 8.0 * g.Tpad;
      TnPlan tp;
      plan_tn(d, g, tp);
      const double wgs = 64.0 * tp.m_tiles * tp.n_tiles * tp.nsplit;
      const double gemm = 16.0 * K * N / (125e12 * fill(wgs, 256.0)), traffic = 32.0 * (K + N) / 4.7e12;
      const double wino = vpad * (36.0 * (K + N) / 4.9e12 + (gemm > traffic ? gemm : traffic));
      // the direct weight gradient splits over voxel chunks, so it keeps the chip full down to ~16k voxels
      const double dfill = vox >= 16384.0 ? 1.0 : vox / 16384.0;
      if (wino < 0.85 * direct / dfill) { best = wino; pick = 1; }
    }
  }
  if (w2 && d->D >= 8 && vox >= 65536.0) {
    const double w2d = vox * 54.0 * K * N / W2D_WGRAD_RATE;     // direct-equivalent rate, measured
    if (w2d < 0.92 * direct && w2d < best) { best = w2d; pick = 2; }
  }
  return pick;
}

extern "C" int dram_wino_pack_weight(const float* w, float* uf, float* ub, int Cout, int Cin, dram_stream_t stream) {
  if (!w || (!uf && !ub) || Cout < 1 || Cin < 1) return DRAM_ERR_BAD_ARG;
  const long n = (long)Cout * Cin;
  hipLaunchKernelGGL(wino_weight_kernel, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream, w, uf,
                     ub, Cout, Cin);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_wino_num_stat_rows(const DramConvDesc* d) {
  if (!dram_wino_applicable(d)) return DRAM_ERR_UNSUPPORTED;
  const WinoGeom g = make_geom(d);
  return (g.T + WINO_TPB - 1) / WINO_TPB;
}

/* pass: 0 forward, 1 data gradient, 2 weight gradient */
extern "C" size_t dram_wino_workspace(const DramConvDesc* d, int pass) {
  if (!dram_wino_applicable(d) || pass < 0 || pass > 2) return 0;
  const WinoGeom g = make_geom(d);
  size_t n = (size_t)64 * g.Tpad * ((size_t)d->Cin + d->Cout);
  if (pass == 2) {
    TnPlan p;
    if (!plan_tn(d, g, p)) return 0;
    n += (size_t)p.nsplit * 64 * d->Cout * d->Cin;
  }
  return n * sizeof(float);
}

extern "C" int dram_wino_conv3d_fwd(const float* x, const float* uf, const float* bias, float* y, float* stats_partial,
                                    float* v_keep, const DramConvDesc* d, void* workspace, size_t workspace_bytes,
                                    dram_stream_t stream) {
  if (!x || !uf || !y) return DRAM_ERR_BAD_ARG;
  if (!dram_wino_applicable(d)) return DRAM_ERR_UNSUPPORTED;
  return run_conv(x, uf, bias, nullptr, nullptr, y, stats_partial, v_keep, d, d->Cin, d->Cout, workspace,
                  workspace_bytes, (hipStream_t)stream);
}

extern "C" size_t dram_wino_v_elems(const DramConvDesc* d) {
  if (!dram_wino_applicable(d)) return 0;
  return (size_t)64 * make_geom(d).Tpad * (size_t)d->Cin;
}

extern "C" int dram_wino_conv3d_bwd_data(const float* dy, const float* ub, float* dx, const float* add,
                                         const float* gate, const DramConvDesc* d, void* workspace,
                                         size_t workspace_bytes, dram_stream_t stream) {
  if (!dy || !ub || !dx || (gate && !add)) return DRAM_ERR_BAD_ARG;
  if (!dram_wino_applicable(d)) return DRAM_ERR_UNSUPPORTED;
  return run_conv(dy, ub, nullptr, add, gate, dx, nullptr, nullptr, d, d->Cout, d->Cin, workspace, workspace_bytes,
                  (hipStream_t)stream);
}

extern "C" int dram_wino_conv3d_bwd_weight(const float* x, const float* v_cache, const float* dy, float* dw,
                                           const DramConvDesc* d, void* workspace, size_t workspace_bytes,
                                           dram_stream_t stream) {
  if ((!x && !v_cache) || !dy || !dw) return DRAM_ERR_BAD_ARG;
  if (!dram_wino_applicable(d)) return DRAM_ERR_UNSUPPORTED;
  const WinoGeom g = make_geom(d);
  TnPlan p;
  if (!plan_tn(d, g, p)) return DRAM_ERR_UNSUPPORTED;
  const size_t need = dram_wino_workspace(d, 2);
  if (!workspace || workspace_bytes < need) return DRAM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  float* V = (float*)workspace;                               // [64][Tpad][Cin]
  float* Dh = V + (size_t)64 * g.Tpad * d->Cin;               // [64][Tpad][Cout]
  float* slab = Dh + (size_t)64 * g.Tpad * d->Cout;           // [nsplit][64][Cout][Cin]
  if (v_cache) V = const_cast<float*>(v_cache);
  else {
    hipLaunchKernelGGL((wino_in_kernel<0>), dim3(grid_for((long)g.Tpad * (d->Cin / 64))), dim3(256), 0, s, x, V, g,
                       d->Cin);
    DRAM_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL((wino_in_kernel<1>), dim3(grid_for((long)g.Tpad * (d->Cout / 64))), dim3(256), 0, s, dy, Dh, g,
                     d->Cout);
  DRAM_LAUNCH_CHECK();
  const int nblk = 64 * p.nsplit * p.m_tiles * p.n_tiles;
#define WTN(WM_, MI_, NJ_)                                                                                       \
  hipLaunchKernelGGL((wino_gemm_tn_kernel<WM_, MI_, NJ_>), dim3(nblk), dim3(512), 0, s, Dh, V, slab, g.Tpad, d->Cout, \
                     d->Cin, p.m_tiles, p.n_tiles, p.nsplit, p.kper, nblk)
  if (p.bm == 256 && p.bn == 256) WTN(4, 2, 4);
  else if (p.bm == 256 && p.bn == 128) WTN(4, 2, 2);
  else if (p.bm == 256 && p.bn == 64) WTN(4, 2, 1);
  else if (p.bm == 128 && p.bn == 256) WTN(2, 2, 2);
  else if (p.bm == 128 && p.bn == 128) WTN(2, 2, 1);
  else if (p.bm == 64 && p.bn == 256) WTN(2, 1, 2);
  else if (p.bm == 64 && p.bn == 128) WTN(2, 1, 1);
  else return DRAM_ERR_UNSUPPORTED;
#undef WTN
  DRAM_LAUNCH_CHECK();
  const long n = (long)d->Cout * d->Cin;
  hipLaunchKernelGGL(wino_wgrad_out_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, s, slab, dw, d->Cout,
                     d->Cin, p.nsplit);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
