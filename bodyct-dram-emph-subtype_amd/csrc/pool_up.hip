// pool_up.hip -- MaxPool3d(3,2,1), trilinear x2 up-projection + crop/concat, and the
// predict-time up-projection to the scan grid.
// Replaces nn.MaxPool3d (reference med3d.py:206/:275), nn.Upsample(trilinear,
// align_corners=True) + crop_concat_5d (med3d.py:83-87, :39-48), F.interpolate at
// models.py:438-441, and their autograd backward.  HBM-bound; float4 = 4 channels/lane.
#include <stdlib.h>
#include <string.h>
#include <initializer_list>
#include "common.h"

namespace {

// one-shot blocks (common.h, ew_blocks): each loop below then runs once per thread
inline int ew_grid(long total) { return ew_blocks(total, 256, 8192); }

// ------------------------------------------------------------------ max pool
// A thread owns VW channels of one voxel (VW * sizeof(T) = 16 B where the channel count allows: common.h, fvec).
template <typename T, int VW>
__global__ void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                   uint8_t* __restrict__ amax, int D, int H, int W, int C, int Do, int Ho, int Wo,
                                   long total) {
  const int Q = C / VW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    long v = i / Q;
    const int xo = (int)(v % Wo); v /= Wo;
    const int yo = (int)(v % Ho); v /= Ho;
    const int zo = (int)(v % Do);
    const long b = v / Do;
    fvec<VW> m;
    int am[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) { m.v[k] = -INFINITY; am[k] = 0; }
    bool first = true;
    // scan order kd, kh, kw; strict '>' keeps the first maximum (ATen max_pool3d semantics).  The nine taps of a
    // plane are loaded together (clamped coordinates, validity flags) -- a guarded load per tap is waited for
    // one at a time
    for (int kz = 0; kz < 3; ++kz) {
      const int zi = 2 * zo - 1 + kz;
      if (zi < 0 || zi >= D) continue;
      fvec<VW> t[3][3];
      bool ok[3][3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yi = 2 * yo - 1 + ky, xi = 2 * xo - 1 + kx;
          ok[ky][kx] = (yi >= 0) & (yi < H) & (xi >= 0) & (xi < W);
          const int yc = yi < 0 ? 0 : (yi >= H ? H - 1 : yi), xc = xi < 0 ? 0 : (xi >= W ? W - 1 : xi);
          t[ky][kx] = ldv<T, VW>(x, (((b * D + zi) * H + yc) * W + xc) * (long)C + VW * q);
        }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          if (!ok[ky][kx]) continue;
          const int tap = (kz * 3 + ky) * 3 + kx;
#pragma unroll
          for (int k = 0; k < VW; ++k) {
            const float tv = t[ky][kx].v[k];
            if (first || tv > m.v[k] || tv != tv) { m.v[k] = tv; am[k] = tap; }
          }
          first = false;
        }
    }
    stv<T, VW>(y, VW * i, m);
#pragma unroll
    for (int k = 0; k < VW; k += 4)
      reinterpret_cast<uchar4*>(amax)[(VW / 4) * i + (k >> 2)] =
          make_uchar4((unsigned char)am[k], (unsigned char)am[k + 1], (unsigned char)am[k + 2], (unsigned char)am[k + 3]);
  }
}

// Stem: BatchNorm-apply + ReLU + max-pool (+ argmax) in ONE pass over the pre-BN convolution output (reference
// med3d.py:272-275: bn1, relu, maxpool).  A thread computes z = relu(y * scale + shift) for the 27 taps of its pooling
// window (the expression and the storage rounding of bn_apply_kernel, so z, the pooled values and the taps are bit-
// identical to the two-pass form), takes the window maximum, and WRITES z for the 2x2x2 block of inputs it owns
// (taps k in {1, 2} per axis: inputs 2o, 2o + 1) -- z is still needed (skip connection of us2, ReLU mask of the
// backward pass); what the fusion saves is the second read of that 537-MB tensor.
template <typename T, int VW>
__global__ void bn_maxpool_fwd_kernel(const T* __restrict__ yin, const float* __restrict__ scale,
                                      const float* __restrict__ shift, T* __restrict__ z, T* __restrict__ pooled,
                                      uint8_t* __restrict__ amax, int D, int H, int W, int C, int Do, int Ho, int Wo,
                                      long total) {
  const int Q = C / VW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    long v = i / Q;
    const int xo = (int)(v % Wo); v /= Wo;
    const int yo = (int)(v % Ho); v /= Ho;
    const int zo = (int)(v % Do);
    const long b = v / Do;
    float sc[VW], sh[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) { sc[k] = scale[VW * q + k]; sh[k] = shift[VW * q + k]; }
    fvec<VW> m;
    int am[VW];
#pragma unroll
    for (int k = 0; k < VW; ++k) { m.v[k] = -INFINITY; am[k] = 0; }
    bool first = true;
    for (int kz = 0; kz < 3; ++kz) {
      const int zi = 2 * zo - 1 + kz;
      if (zi < 0 || zi >= D) continue;
      fvec<VW> t[3][3];
      bool ok[3][3];
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int yi = 2 * yo - 1 + ky, xi = 2 * xo - 1 + kx;
          ok[ky][kx] = (yi >= 0) & (yi < H) & (xi >= 0) & (xi < W);
          const int yc = yi < 0 ? 0 : (yi >= H ? H - 1 : yi), xc = xi < 0 ? 0 : (xi >= W ? W - 1 : xi);
          t[ky][kx] = ldv<T, VW>(yin, (((b * D + zi) * H + yc) * W + xc) * (long)C + VW * q);
        }
#pragma unroll
      for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          if (!ok[ky][kx]) continue;
          const int tap = (kz * 3 + ky) * 3 + kx;
          fvec<VW> a;
#pragma unroll
          for (int k = 0; k < VW; ++k) {
            float tv = fmaxf(__builtin_fmaf(t[ky][kx].v[k], sc[k], sh[k]), 0.f);
            if (sizeof(T) == 2) tv = bf16_to_f32(f32_to_bf16(tv));        // the value the two-pass form stores and re-reads
            a.v[k] = tv;
            if (first || tv > m.v[k] || tv != tv) { m.v[k] = tv; am[k] = tap; }
          }
          first = false;
          if (kz >= 1 && ky >= 1 && kx >= 1) {                            // an input this window owns
            const int yi = 2 * yo - 1 + ky, xi = 2 * xo - 1 + kx;
            stv<T, VW>(z, (((b * D + zi) * H + yi) * W + xi) * (long)C + VW * q, a);
          }
        }
    }
    stv<T, VW>(pooled, VW * i, m);
#pragma unroll
    for (int k = 0; k < VW; k += 4)
      reinterpret_cast<uchar4*>(amax)[(VW / 4) * i + (k >> 2)] =
          make_uchar4((unsigned char)am[k], (unsigned char)am[k + 1], (unsigned char)am[k + 2], (unsigned char)am[k + 3]);
  }
}

template <typename T, int VW>
__global__ void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ amax,
                                   const T* __restrict__ add, int add_stride, T* __restrict__ dx, int D, int H,
                                   int W, int C, int Do, int Ho, int Wo, long total) {
  const int Q = C / VW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    long v = i / Q;
    const int xi = (int)(v % W); v /= W;
    const int yi = (int)(v % H); v /= H;
    const int zi = (int)(v % D);
    const long b = v / D;
    // add: a tensor shaped like dx, or a channel slice of a wider one (add_stride elements per voxel)
    fvec<VW> s;
    if (add) s = ldv<T, VW>(add, (i / Q) * (long)add_stride + VW * q);
    else {
#pragma unroll
      for (int k = 0; k < VW; ++k) s.v[k] = 0.f;
    }
    // windows containing zi: zo with 2zo-1 <= zi <= 2zo+1
    const int zlo = zi >> 1, zhi = (zi + 1) >> 1;  // ceil((zi-1)/2) == zi>>1 for zi>=0
    const int ylo = yi >> 1, yhi = (yi + 1) >> 1;
    const int xlo = xi >> 1, xhi = (xi + 1) >> 1;
    // the (up to) 2 x 2 x 2 windows containing this voxel: all eight (argmax, dy) pairs are loaded together with
    // clamped window indices, invalid ones are skipped afterwards -- in the original scan order
    uchar4 am[2][2][2][VW / 4];
    fvec<VW> gg[2][2][2];
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
        for (int dx_ = 0; dx_ < 2; ++dx_) {
          const int zo = min(zlo + dz, Do - 1), yo = min(ylo + dy_, Ho - 1), xo = min(xlo + dx_, Wo - 1);
          const long o = ((((b * Do + zo) * Ho + yo) * Wo + xo) * (long)Q + q);
#pragma unroll
          for (int k = 0; k < VW / 4; ++k) am[dz][dy_][dx_][k] = reinterpret_cast<const uchar4*>(amax)[(VW / 4) * o + k];
          gg[dz][dy_][dx_] = ldv<T, VW>(dy, VW * o);
        }
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
        for (int dx_ = 0; dx_ < 2; ++dx_) {
          const int zo = zlo + dz, yo = ylo + dy_, xo = xlo + dx_;
          if (zo > zhi || zo >= Do || yo > yhi || yo >= Ho || xo > xhi || xo >= Wo) continue;
          const int kz = zi - (2 * zo - 1), ky = yi - (2 * yo - 1), kx = xi - (2 * xo - 1);
          const unsigned char tap = (unsigned char)((kz * 3 + ky) * 3 + kx);
#pragma unroll
          for (int k = 0; k < VW / 4; ++k) {
            const uchar4 a = am[dz][dy_][dx_][k];
            const fvec<VW>& g = gg[dz][dy_][dx_];
            if (a.x == tap) s.v[4 * k] += g.v[4 * k];
            if (a.y == tap) s.v[4 * k + 1] += g.v[4 * k + 1];
            if (a.z == tap) s.v[4 * k + 2] += g.v[4 * k + 2];
            if (a.w == tap) s.v[4 * k + 3] += g.v[4 * k + 3];
          }
        }
    stv<T, VW>(dx, VW * i, s);
  }
}

// The same gradient with one thread per 2 x 2 x 2 block of input voxels: the block's voxels lie in the (up to) eight
// windows {a, a + 1}^3 only -- per axis an even coordinate 2a belongs to window a (tap 1), an odd one 2a + 1 to
// windows a (tap 2) and a + 1 (tap 0) -- so the eight (argmax, dy) pairs are loaded ONCE per block instead of once
// per voxel (16 + 8 loads per 8 voxels instead of 136; the per-voxel kernel ran at 2.9 TB/s on its gathers).
// Contributions are added in the per-voxel kernel's window order: bit-identical results.
template <typename T, int VW>
__global__ void maxpool_bwd_blk_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ amax,
                                       const T* __restrict__ add, int add_stride, T* __restrict__ dx, int D, int H,
                                       int W, int C, int Do, int Ho, int Wo, long total) {
  const int Q = C / VW;
  const int Db = (D + 1) >> 1, Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    long v = i / Q;
    const int cx = (int)(v % Wb); v /= Wb;
    const int by = (int)(v % Hb); v /= Hb;
    const int az = (int)(v % Db);
    const long b = v / Db;
    uchar4 am[2][2][2][VW / 4];
    fvec<VW> gg[2][2][2];
#pragma unroll
    for (int wz = 0; wz < 2; ++wz)
#pragma unroll
      for (int wy = 0; wy < 2; ++wy)
#pragma unroll
        for (int wx = 0; wx < 2; ++wx) {     // clamped indices: windows past the pooled grid are never used below
          const int zo = min(az + wz, Do - 1), yo = min(by + wy, Ho - 1), xo = min(cx + wx, Wo - 1);
          const long o = ((((b * Do + zo) * Ho + yo) * Wo + xo) * (long)Q + q);
#pragma unroll
          for (int k = 0; k < VW / 4; ++k) am[wz][wy][wx][k] = reinterpret_cast<const uchar4*>(amax)[(VW / 4) * o + k];
          gg[wz][wy][wx] = ldv<T, VW>(dy, VW * o);
        }
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
      for (int dy_ = 0; dy_ < 2; ++dy_)
#pragma unroll
        for (int dx_ = 0; dx_ < 2; ++dx_) {
          const int z = 2 * az + dz, y = 2 * by + dy_, x = 2 * cx + dx_;
          if (z >= D || y >= H || x >= W) continue;
          const long vox = ((b * D + z) * H + y) * (long)W + x;
          fvec<VW> s;
          if (add) s = ldv<T, VW>(add, vox * add_stride + VW * q);
          else {
#pragma unroll
            for (int k = 0; k < VW; ++k) s.v[k] = 0.f;
          }
#pragma unroll
          for (int wz = 0; wz <= dz; ++wz)
#pragma unroll
            for (int wy = 0; wy <= dy_; ++wy)
#pragma unroll
              for (int wx = 0; wx <= dx_; ++wx) {
                if (az + wz >= Do || by + wy >= Ho || cx + wx >= Wo) continue;
                // tap of voxel coordinate 2a + d in window a + w: d + 1 - 2 w  (d = 0: 1; d = 1: 2 in window a, 0 in a + 1)
                const unsigned char tap = (unsigned char)(((dz + 1 - 2 * wz) * 3 + (dy_ + 1 - 2 * wy)) * 3 + (dx_ + 1 - 2 * wx));
#pragma unroll
                for (int k = 0; k < VW / 4; ++k) {
                  const uchar4 a = am[wz][wy][wx][k];
                  const fvec<VW>& g = gg[wz][wy][wx];
                  if (a.x == tap) s.v[4 * k] += g.v[4 * k];
                  if (a.y == tap) s.v[4 * k + 1] += g.v[4 * k + 1];
                  if (a.z == tap) s.v[4 * k + 2] += g.v[4 * k + 2];
                  if (a.w == tap) s.v[4 * k + 3] += g.v[4 * k + 3];
                }
              }
          stv<T, VW>(dx, vox * C + VW * q, s);
        }
  }
}

// ------------------------------------------------------------------ trilinear helpers: lin_src (common.h)

template <typename T, int VW>
__global__ void upcat_fwd_kernel(const T* __restrict__ src, const T* __restrict__ skip,
                                 T* __restrict__ cat, int Ds, int Hs, int Ws, int Cu, int Dk, int Hk, int Wk,
                                 int Ck, int oz, int oy, int ox, float sz, float sy, float sx, long total) {
  const int Do = 2 * Ds, Ho = 2 * Hs, Wo = 2 * Ws;
  const int Ct = Cu + Ck, Q = Ct / VW, Qu = Cu / VW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    long v = i / Q;
    const int xo = (int)(v % Wo); v /= Wo;
    const int yo = (int)(v % Ho); v /= Ho;
    const int zo = (int)(v % Do);
    const long b = v / Do;
    fvec<VW> o;
    if (q < Qu) {
      int z0, z1, y0, y1, x0, x1;
      float wz0, wz1, wy0, wy1, wx0, wx1;
      lin_src(zo, sz, Ds, z0, z1, wz0, wz1);
      lin_src(yo, sy, Hs, y0, y1, wy0, wy1);
      lin_src(xo, sx, Ws, x0, x1, wx0, wx1);
#pragma unroll
      for (int k = 0; k < VW; ++k) o.v[k] = 0.f;
#define UP_ACC(zz, yy, xx, ww)                                                                                   \
  {                                                                                                              \
    const fvec<VW> t = ldv<T, VW>(src, (((b * Ds + zz) * Hs + yy) * Ws + xx) * (long)Cu + VW * q);               \
    const float w_ = (ww);                                                                                       \
    _Pragma("unroll") for (int k = 0; k < VW; ++k) o.v[k] += w_ * t.v[k];                                        \
  }
      // same association as ATen: w0z*(w0y*(w0x*a + w1x*b) + ...) is not reproduced bit-wise; fp32 tolerance
      UP_ACC(z0, y0, x0, wz0 * wy0 * wx0) UP_ACC(z0, y0, x1, wz0 * wy0 * wx1)
      UP_ACC(z0, y1, x0, wz0 * wy1 * wx0) UP_ACC(z0, y1, x1, wz0 * wy1 * wx1)
      UP_ACC(z1, y0, x0, wz1 * wy0 * wx0) UP_ACC(z1, y0, x1, wz1 * wy0 * wx1)
      UP_ACC(z1, y1, x0, wz1 * wy1 * wx0) UP_ACC(z1, y1, x1, wz1 * wy1 * wx1)
#undef UP_ACC
    } else {
      o = ldv<T, VW>(skip, (((b * Dk + zo + oz) * Hk + yo + oy) * Wk + xo + ox) * (long)Ck + VW * (q - Qu));
    }
    stv<T, VW>(cat, VW * i, o);
  }
}

// transposed trilinear as a gather: each source voxel collects from the destination voxels
// whose interpolation stencil touches it (deterministic, no atomics).
// LDS-tiled form of upcat_fwd for Cu % 64 == 0: a workgroup produces an 8 x 8 x 8 block of output voxels.  The (at
// most) 6 x 6 x 6 source voxels it interpolates from (7 * scale < 3.5, so the upper neighbour of the last output is at
// most 5 past the first output's floor) are loaded once per 64-channel block into LDS instead of
// being gathered eight times per output from L2 / MALL (8 x the output bytes: the untiled kernel ran at 2.5 TB/s).
// Same corner order and weights as upcat_fwd_kernel.
template <typename T, int VW>
__global__ __launch_bounds__(256) void upcat_fwd_tiled_kernel(const T* __restrict__ src, const T* __restrict__ skip,
                                                              T* __restrict__ cat, int Ds, int Hs, int Ws, int Cu,
                                                              int Dk, int Hk, int Wk, int Ck, int oz, int oy, int ox,
                                                              float sz, float sy, float sx, int tz_n, int ty_n,
                                                              int tx_n) {
  constexpr int LPV = 64 / VW;               // lanes per voxel of a 64-channel block
  constexpr int SLOTS = 256 / LPV;           // voxels per pass
  __shared__ float4 tile[216 * 16];          // [6][6][6] voxels x 64 channels, fp32 (converted once at the fill)
  const int Do = 2 * Ds, Ho = 2 * Hs, Wo = 2 * Ws;
  const int Ct = Cu + Ck, Qk = Ck / VW;
  int r = blockIdx.x;
  const int txi = r % tx_n; r /= tx_n;
  const int tyi = r % ty_n; r /= ty_n;
  const int tzi = r % tz_n;
  const long b = r / tz_n;
  const int z0 = tzi * 8, y0 = tyi * 8, x0 = txi * 8;
  // first source index of the block along each axis (lin_src's floor of the block's first output)
  int zb, yb, xb, d1;
  float w0, w1;
  lin_src(z0, sz, Ds, zb, d1, w0, w1);
  lin_src(y0, sy, Hs, yb, d1, w0, w1);
  lin_src(x0, sx, Ws, xb, d1, w0, w1);
  const int tid = threadIdx.x, q = tid % LPV, vs = tid / LPV;     // channel group, voxel slot
  for (int cb = 0; cb < Cu; cb += 64) {
    __syncthreads();                       // previous channel block consumed
    for (int e = tid; e < 216 * LPV; e += 256) {
      const int qq = e % LPV, v = e / LPV;
      const int lx = v % 6, ly = (v / 6) % 6, lz = v / 36;
      const int zz = min(zb + lz, Ds - 1), yy = min(yb + ly, Hs - 1), xx = min(xb + lx, Ws - 1);
      const fvec<VW> t = ldv<T, VW>(src, (((b * Ds + zz) * Hs + yy) * Ws + xx) * (long)Cu + cb + VW * qq);
#pragma unroll
      for (int k = 0; k < VW; k += 4)
        tile[v * 16 + (VW / 4) * qq + (k >> 2)] = make_float4(t.v[k], t.v[k + 1], t.v[k + 2], t.v[k + 3]);
    }
    __syncthreads();
    for (int v = vs; v < 512; v += SLOTS) {
      const int zo = z0 + (v >> 6), yo = y0 + ((v >> 3) & 7), xo = x0 + (v & 7);
      if (zo >= Do || yo >= Ho || xo >= Wo) continue;
      int za, zc, ya, yc, xa, xc;
      float wz0, wz1, wy0, wy1, wx0, wx1;
      lin_src(zo, sz, Ds, za, zc, wz0, wz1);
      lin_src(yo, sy, Hs, ya, yc, wy0, wy1);
      lin_src(xo, sx, Ws, xa, xc, wx0, wx1);
      za -= zb; zc -= zb; ya -= yb; yc -= yb; xa -= xb; xc -= xb;
      fvec<VW> o;
#pragma unroll
      for (int k = 0; k < VW; ++k) o.v[k] = 0.f;
#define UPT_ACC(zz, yy, xx, ww)                                                              \
  {                                                                                          \
    const float w_ = (ww);                                                                   \
    _Pragma("unroll") for (int k = 0; k < VW; k += 4) {                                      \
      const float4 t = tile[(((zz) * 6 + (yy)) * 6 + (xx)) * 16 + (VW / 4) * q + (k >> 2)];  \
      o.v[k] += w_ * t.x; o.v[k + 1] += w_ * t.y; o.v[k + 2] += w_ * t.z; o.v[k + 3] += w_ * t.w; \
    }                                                                                        \
  }
      UPT_ACC(za, ya, xa, wz0 * wy0 * wx0) UPT_ACC(za, ya, xc, wz0 * wy0 * wx1)
      UPT_ACC(za, yc, xa, wz0 * wy1 * wx0) UPT_ACC(za, yc, xc, wz0 * wy1 * wx1)
      UPT_ACC(zc, ya, xa, wz1 * wy0 * wx0) UPT_ACC(zc, ya, xc, wz1 * wy0 * wx1)
      UPT_ACC(zc, yc, xa, wz1 * wy1 * wx0) UPT_ACC(zc, yc, xc, wz1 * wy1 * wx1)
#undef UPT_ACC
      const long vox = ((b * Do + zo) * Ho + yo) * (long)Wo + xo;
      stv<T, VW>(cat, vox * Ct + cb + VW * q, o);
    }
  }
  // centre-cropped skip connection -> channels Cu .. Ct - 1
  for (int e = tid; e < 512 * Qk; e += 256) {
    const int qq = e % Qk, v = e / Qk;
    const int zo = z0 + (v >> 6), yo = y0 + ((v >> 3) & 7), xo = x0 + (v & 7);
    if (zo >= Do || yo >= Ho || xo >= Wo) continue;
    const long vox = ((b * Do + zo) * Ho + yo) * (long)Wo + xo;
    stv<T, VW>(cat, vox * Ct + Cu + VW * qq,
               ldv<T, VW>(skip, (((b * Dk + zo + oz) * Hk + yo + oy) * Wk + xo + ox) * (long)Ck + VW * qq));
  }
}

__device__ __forceinline__ void dst_range(int s, float scale, int out, int& lo, int& hi) {
  // destinations d with floor(scale*d) in {s-1, s}: conservative bounds, exact test in the loop
  if (scale <= 0.f) { lo = 0; hi = out - 1; return; }
  const float inv = 1.f / scale;
  lo = (int)floorf((float)(s - 1) * inv) - 1;
  hi = (int)ceilf((float)(s + 1) * inv) + 1;
  if (lo < 0) lo = 0;
  if (hi > out - 1) hi = out - 1;
}

// per-axis table of the (at most 6) destinations that read source s, with their weights.  Destinations d with
// floor(scale * d) in {s - 1, s} span 2 / scale < 4.1 indices for a x2 up-sampling (scale = (in-1)/(2in-1) < 1/2), so a
// window of 6 starting at floor((s - 1) / scale) - (rounding slack) covers them; the rest of the window gets weight 0.
struct UpAxis {
  int lo;
  float w[6];
};
__device__ __forceinline__ UpAxis up_axis(int s, float scale, int in, int out) {
  UpAxis t;
  int lo = scale > 0.f ? (int)floorf((float)(s - 1) / scale) - 1 : 0;
  if (lo < 0) lo = 0;
  t.lo = lo;
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    const int d = lo + j;
    int a0, a1;
    float u0, u1;
    lin_src(d, scale, in, a0, a1, u0, u1);
    const float w = (a0 == s ? u0 : 0.f) + (a1 == s ? u1 : 0.f);
    t.w[j] = d < out ? w : 0.f;
  }
  return t;
}

// Transposed x2 trilinear (align_corners) as a gather: one thread = 4 channels of one SOURCE voxel.  The weights are
// separable and channel-independent: three 6-entry tables (18 lin_src evaluations; the first version re-derived them
// inside a 7x7x7 candidate loop with three levels of data-dependent `continue`, one load in flight: 1.3 TB/s), then
// rows of 6 predicated loads issued together.
template <typename T, int VW>
__global__ void upcat_bwd_src_kernel(const T* __restrict__ dcat, T* __restrict__ dsrc, int Ds, int Hs, int Ws,
                                     int Cu, int Ct, float sz, float sy, float sx, long total) {
  const int Do = 2 * Ds, Ho = 2 * Hs, Wo = 2 * Ws;
  const int Qu = Cu / VW;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Qu);
    long v = i / Qu;
    const int xs = (int)(v % Ws); v /= Ws;
    const int ys = (int)(v % Hs); v /= Hs;
    const int zs = (int)(v % Ds);
    const long b = v / Ds;
    const UpAxis tz = up_axis(zs, sz, Ds, Do), ty = up_axis(ys, sy, Hs, Ho), tx = up_axis(xs, sx, Ws, Wo);
    fvec<VW> acc;
#pragma unroll
    for (int k = 0; k < VW; ++k) acc.v[k] = 0.f;
    // (both loops fully unrolled: indexed with a loop variable the weight tables were moved to LDS by the compiler --
    // 56 KB per workgroup, two workgroups per CU for a kernel that lives on loads in flight)
#pragma unroll
    for (int jz = 0; jz < 6; ++jz) {
      if (tz.w[jz] == 0.f) continue;
#pragma unroll
      for (int jy = 0; jy < 6; ++jy) {
        const float wzy = tz.w[jz] * ty.w[jy];
        if (wzy == 0.f) continue;
        const T* row = dcat + (((b * Do + tz.lo + jz) * Ho + ty.lo + jy) * (long)Wo + tx.lo) * (long)Ct + VW * q;
        fvec<VW> g[6];
#pragma unroll
        for (int jx = 0; jx < 6; ++jx)      // clamped address + zero weight instead of a branch per load
          g[jx] = ldv<T, VW>(row, (long)(tx.w[jx] != 0.f ? jx : 0) * Ct);
#pragma unroll
        for (int jx = 0; jx < 6; ++jx) {
          const float w_ = wzy * tx.w[jx];
#pragma unroll
          for (int k = 0; k < VW; ++k) acc.v[k] = __builtin_fmaf(w_, g[jx].v[k], acc.v[k]);   // (explicit: every instantiation contracts alike)
        }
      }
    }
    stv<T, VW>(dsrc, VW * i, acc);
  }
}

template <typename T, int VW>
__global__ void upcat_bwd_skip_kernel(const T* __restrict__ dcat, T* __restrict__ dskip, int Do, int Ho,
                                      int Wo, int Cu, int Dk, int Hk, int Wk, int Ck, int oz, int oy, int ox,
                                      long total) {
  const int Qk = Ck / VW;
  const int Ct = Cu + Ck;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Qk);
    long v = i / Qk;
    const int xk = (int)(v % Wk); v /= Wk;
    const int yk = (int)(v % Hk); v /= Hk;
    const int zk = (int)(v % Dk);
    const long b = v / Dk;
    const int zo = zk - oz, yo = yk - oy, xo = xk - ox;
    fvec<VW> o;
#pragma unroll
    for (int k = 0; k < VW; ++k) o.v[k] = 0.f;
    if (zo >= 0 && zo < Do && yo >= 0 && yo < Ho && xo >= 0 && xo < Wo)
      o = ldv<T, VW>(dcat, (((b * Do + zo) * Ho + yo) * Wo + xo) * (long)Ct + Cu + VW * q);
    stv<T, VW>(dskip, VW * i, o);
  }
}

// predict-time: out = trilinear(dense -> (Do,Ho,Wo), align_corners) * ess ; per-block sums
__global__ __launch_bounds__(256) void upproject_kernel(const float* __restrict__ dense, const float* __restrict__ ess,
                                                        float* __restrict__ out, float* __restrict__ partial, int D,
                                                        int H, int W, int Do, int Ho, int Wo, float sz, float sy,
                                                        float sx, long vps, int nblk) {
  __shared__ float sm[4];
  const int b = blockIdx.y;
  float s = 0.f;
  for (long v = blockIdx.x * (long)blockDim.x + threadIdx.x; v < vps; v += (long)gridDim.x * blockDim.x) {
    long r = v;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho);
    const int zo = (int)(r / Ho);
    int z0, z1, y0, y1, x0, x1;
    float wz0, wz1, wy0, wy1, wx0, wx1;
    lin_src(zo, sz, D, z0, z1, wz0, wz1);
    lin_src(yo, sy, H, y0, y1, wy0, wy1);
    lin_src(xo, sx, W, x0, x1, wx0, wx1);
    const float* p = dense + (long)b * D * H * W;
#define DV(zz, yy, xx) p[((long)(zz) * H + (yy)) * W + (xx)]
    const float val = wz0 * (wy0 * (wx0 * DV(z0, y0, x0) + wx1 * DV(z0, y0, x1)) + wy1 * (wx0 * DV(z0, y1, x0) + wx1 * DV(z0, y1, x1))) +
                      wz1 * (wy0 * (wx0 * DV(z1, y0, x0) + wx1 * DV(z1, y0, x1)) + wy1 * (wx0 * DV(z1, y1, x0) + wx1 * DV(z1, y1, x1)));
#undef DV
    const float o = val * ess[(long)b * vps + v];
    out[(long)b * vps + v] = o;
    s += o;
  }
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[(long)b * nblk + blockIdx.x] = sm[0] + sm[1] + sm[2] + sm[3];
}

inline float ac_scale(int in, int out) { return out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f; }

}  // namespace

// channels per thread: 8 for bf16 storage when every channel count / stride involved is a multiple of 8 (16-B
// vectors), else 4.  DRAM_POOL_VW=4 forces the narrow form (A/B).
static inline bool pool_wide(size_t elem, std::initializer_list<long> counts) {
  static const bool narrow = tune_env("DRAM_POOL_VW") && atoi(tune_env("DRAM_POOL_VW")) == 4;
  if (elem != 2 || narrow) return false;
  for (long c : counts)
    if (c & 7) return false;
  return true;
}
#define POOL_LAUNCH(WIDE_, KERNEL_, GRID_, ...)                                                                       \
  do {                                                                                                                \
    if constexpr (sizeof(T) == 2) {                                                                                   \
      if (WIDE_) hipLaunchKernelGGL((KERNEL_<T, 8>), dim3(GRID_), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);     \
      else hipLaunchKernelGGL((KERNEL_<T, 4>), dim3(GRID_), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);           \
    } else {                                                                                                          \
      hipLaunchKernelGGL((KERNEL_<T, 4>), dim3(GRID_), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__);                \
    }                                                                                                                 \
  } while (0)

template <typename T>
static int maxpool_fwd_impl(const T* x, T* y, uint8_t* argmax, int B, int D, int H, int W, int C, dram_stream_t stream) {
  if (!x || !y || !argmax || B < 1 || D < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return DRAM_ERR_BAD_ARG;
  const int Do = (D + 2 - 3) / 2 + 1, Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total4 = (long)B * Do * Ho * Wo * (C >> 2);
  DramProf prof(DRAM_FAM_POOL_UP, 0, 0.0, (double)sizeof(T) * ((double)B * D * H * W * C + 4.0 * total4) + 4.0 * total4,
                (hipStream_t)stream);
  const bool wide = pool_wide(sizeof(T), {C});
  const long total = wide ? total4 / 2 : total4;
  POOL_LAUNCH(wide, maxpool_fwd_kernel, ew_grid(total), x, y, argmax, D, H, W, C, Do, Ho, Wo, total);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
template <typename T>
static int bn_maxpool_fwd_impl(const T* y, const float* scale, const float* shift, T* z, T* pooled, uint8_t* argmax, int B,
                               int D, int H, int W, int C, dram_stream_t stream) {
  if (!y || !scale || !shift || !z || !pooled || !argmax || B < 1 || D < 1 || H < 1 || W < 1 || C < 4 || (C & 3))
    return DRAM_ERR_BAD_ARG;
  const int Do = (D + 2 - 3) / 2 + 1, Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total4 = (long)B * Do * Ho * Wo * (C >> 2);
  DramProf prof(DRAM_FAM_POOL_UP, 6, 0.0, (double)sizeof(T) * (2.0 * (double)B * D * H * W * C + 4.0 * total4) + 4.0 * total4,
                (hipStream_t)stream);
  const bool wide = pool_wide(sizeof(T), {C});
  const long total = wide ? total4 / 2 : total4;
  POOL_LAUNCH(wide, bn_maxpool_fwd_kernel, ew_grid(total), y, scale, shift, z, pooled, argmax, D, H, W, C, Do, Ho, Wo, total);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_bn_maxpool_fwd(const float* y, const float* scale, const float* shift, float* z, float* pooled,
                                   uint8_t* argmax, int B, int D, int H, int W, int C, dram_stream_t stream) {
  return bn_maxpool_fwd_impl<float>(y, scale, shift, z, pooled, argmax, B, D, H, W, C, stream);
}
extern "C" int dram_bn_maxpool_fwd_bf16(const void* y, const float* scale, const float* shift, void* z, void* pooled,
                                        uint8_t* argmax, int B, int D, int H, int W, int C, dram_stream_t stream) {
  return bn_maxpool_fwd_impl<bf16_t>((const bf16_t*)y, scale, shift, (bf16_t*)z, (bf16_t*)pooled, argmax, B, D, H, W, C,
                                     stream);
}

extern "C" int dram_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int B, int D, int H, int W, int C,
                                dram_stream_t stream) {
  return maxpool_fwd_impl<float>(x, y, argmax, B, D, H, W, C, stream);
}
extern "C" int dram_maxpool_fwd_bf16(const void* x, void* y, uint8_t* argmax, int B, int D, int H, int W, int C,
                                     dram_stream_t stream) {
  return maxpool_fwd_impl<bf16_t>((const bf16_t*)x, (bf16_t*)y, argmax, B, D, H, W, C, stream);
}

template <typename T>
static int maxpool_bwd_impl(const T* dy, const uint8_t* argmax, const T* add, int add_stride, T* dx, int B, int D, int H,
                            int W, int C, dram_stream_t stream) {
  if (!dy || !dx || !argmax || B < 1 || D < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return DRAM_ERR_BAD_ARG;
  if (add && (add_stride < C || (add_stride & 3) || ((uintptr_t)add & (4 * sizeof(T) - 1)))) return DRAM_ERR_BAD_ARG;
  const int Do = (D + 2 - 3) / 2 + 1, Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total4 = (long)B * D * H * W * (C >> 2);
  DramProf prof(DRAM_FAM_POOL_UP, 1, 0.0,
                (double)sizeof(T) * 4.0 * total4 * (1.0 + (add ? 1 : 0)) + (1.0 + sizeof(T)) * (double)B * Do * Ho * Wo * C,
                (hipStream_t)stream);
  const bool wide = pool_wide(sizeof(T), {C, add ? add_stride : 0, (add && ((uintptr_t)add & 15)) ? 1 : 0});
  static const bool per_voxel = tune_env("DRAM_POOL_BWD") && !strcmp(tune_env("DRAM_POOL_BWD"), "voxel");     // A/B
  if (!per_voxel) {
    const long totb = (long)B * ((D + 1) / 2) * ((H + 1) / 2) * ((W + 1) / 2) * (C / (wide ? 8 : 4));
    POOL_LAUNCH(wide, maxpool_bwd_blk_kernel, ew_grid(totb), dy, argmax, add, add_stride, dx, D, H, W, C, Do, Ho, Wo, totb);
    DRAM_LAUNCH_CHECK();
    return DRAM_OK;
  }
  const long total = wide ? total4 / 2 : total4;
  POOL_LAUNCH(wide, maxpool_bwd_kernel, ew_grid(total), dy, argmax, add, add_stride, dx, D, H, W, C, Do, Ho, Wo, total);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_maxpool_bwd(const float* dy, const uint8_t* argmax, const float* add, int add_stride, float* dx,
                                int B, int D, int H, int W, int C, dram_stream_t stream) {
  return maxpool_bwd_impl<float>(dy, argmax, add, add_stride, dx, B, D, H, W, C, stream);
}
extern "C" int dram_maxpool_bwd_bf16(const void* dy, const uint8_t* argmax, const void* add, int add_stride, void* dx,
                                     int B, int D, int H, int W, int C, dram_stream_t stream) {
  return maxpool_bwd_impl<bf16_t>((const bf16_t*)dy, argmax, (const bf16_t*)add, add_stride, (bf16_t*)dx, B, D, H, W, C,
                                  stream);
}

template <typename T>
static int upcat_fwd_impl(const T* src, const T* skip, T* cat, int B, int Ds, int Hs, int Ws, int Cu, int Dk, int Hk,
                          int Wk, int Ck, dram_stream_t stream) {
  // (skip == NULL with Ck == 0: the up-sampled tensor alone, cropped to [Dk/2*2 ...] like the concatenation would be --
  // the convolution behind it then takes the skip tensor as its own second source, dram_wino_conv3d_fwd_cat)
  if (!src || !cat || B < 1 || Cu < 4 || (Cu & 3) || (skip ? (Ck < 4 || (Ck & 3)) : Ck != 0)) return DRAM_ERR_BAD_ARG;
  const int Do = 2 * Ds, Ho = 2 * Hs, Wo = 2 * Ws;
  if (Dk < Do || Hk < Ho || Wk < Wo) return DRAM_ERR_BAD_ARG;  // crop_concat_5d assumes t1 <= t2
  const int oz = (Dk - Do + 1) / 2, oy = (Hk - Ho + 1) / 2, ox = (Wk - Wo + 1) / 2;  // ceil((b-a)/2)
  const long total4 = (long)B * Do * Ho * Wo * ((Cu + Ck) >> 2);
  const long tiles = (long)B * ((Do + 7) / 8) * ((Ho + 7) / 8) * ((Wo + 7) / 8);
  DramProf prof(DRAM_FAM_POOL_UP, 2, 0.0,
                (double)sizeof(T) * ((double)B * Ds * Hs * Ws * Cu + (double)B * Do * Ho * Wo * Ck + 4.0 * total4),
                (hipStream_t)stream);
  const bool wide = pool_wide(sizeof(T), {Cu, Ck});
  if (Cu % 64 == 0 && tiles >= 512 && tiles < (1L << 31) && !tune_env("DRAM_UPCAT_UNTILED")) {
    POOL_LAUNCH(wide, upcat_fwd_tiled_kernel, (unsigned)tiles, src, skip, cat, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck, oz, oy, ox,
                ac_scale(Ds, Do), ac_scale(Hs, Ho), ac_scale(Ws, Wo), (Do + 7) / 8, (Ho + 7) / 8, (Wo + 7) / 8);
    DRAM_LAUNCH_CHECK();
    return DRAM_OK;
  }
  const long total = wide ? total4 / 2 : total4;
  POOL_LAUNCH(wide, upcat_fwd_kernel, ew_grid(total), src, skip, cat, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck, oz, oy, ox,
              ac_scale(Ds, Do), ac_scale(Hs, Ho), ac_scale(Ws, Wo), total);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_upcat_fwd(const float* src, const float* skip, float* cat, int B, int Ds, int Hs, int Ws, int Cu,
                              int Dk, int Hk, int Wk, int Ck, dram_stream_t stream) {
  return upcat_fwd_impl<float>(src, skip, cat, B, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck, stream);
}
extern "C" int dram_upcat_fwd_bf16(const void* src, const void* skip, void* cat, int B, int Ds, int Hs, int Ws, int Cu,
                                   int Dk, int Hk, int Wk, int Ck, dram_stream_t stream) {
  return upcat_fwd_impl<bf16_t>((const bf16_t*)src, (const bf16_t*)skip, (bf16_t*)cat, B, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck,
                                stream);
}

template <typename T>
static int upcat_bwd_impl(const T* dcat, T* dsrc, T* dskip, int B, int Ds, int Hs, int Ws, int Cu, int Dk, int Hk, int Wk,
                          int Ck, dram_stream_t stream) {
  if (!dcat || (!dsrc && !dskip) || B < 1 || Cu < 4 || Ck < 4 || (Cu & 3) || (Ck & 3)) return DRAM_ERR_BAD_ARG;
  const int Do = 2 * Ds, Ho = 2 * Hs, Wo = 2 * Ws;
  if (Dk < Do || Hk < Ho || Wk < Wo) return DRAM_ERR_BAD_ARG;
  const int oz = (Dk - Do + 1) / 2, oy = (Hk - Ho + 1) / 2, ox = (Wk - Wo + 1) / 2;
  hipStream_t s = (hipStream_t)stream;
  const bool wide = pool_wide(sizeof(T), {Cu, Ck});
  if (dsrc) {
    const long total4 = (long)B * Ds * Hs * Ws * (Cu >> 2);
    DramProf prof(DRAM_FAM_POOL_UP, 3, 0.0, (double)sizeof(T) * ((double)B * Do * Ho * Wo * Cu + 4.0 * total4), s);
    const long total = wide ? total4 / 2 : total4;
    POOL_LAUNCH(wide, upcat_bwd_src_kernel, ew_grid(total), dcat, dsrc, Ds, Hs, Ws, Cu, Cu + Ck, ac_scale(Ds, Do),
                ac_scale(Hs, Ho), ac_scale(Ws, Wo), total);
    DRAM_LAUNCH_CHECK();
  }
  if (dskip) {
    const long total4 = (long)B * Dk * Hk * Wk * (Ck >> 2);
    DramProf prof(DRAM_FAM_POOL_UP, 4, 0.0, (double)sizeof(T) * ((double)B * Do * Ho * Wo * Ck + 4.0 * total4), s);
    const long total = wide ? total4 / 2 : total4;
    POOL_LAUNCH(wide, upcat_bwd_skip_kernel, ew_grid(total), dcat, dskip, Do, Ho, Wo, Cu, Dk, Hk, Wk, Ck, oz, oy, ox,
                total);
    DRAM_LAUNCH_CHECK();
  }
  return DRAM_OK;
}
extern "C" int dram_upcat_bwd(const float* dcat, float* dsrc, float* dskip, int B, int Ds, int Hs, int Ws, int Cu,
                              int Dk, int Hk, int Wk, int Ck, dram_stream_t stream) {
  return upcat_bwd_impl<float>(dcat, dsrc, dskip, B, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck, stream);
}
extern "C" int dram_upcat_bwd_bf16(const void* dcat, void* dsrc, void* dskip, int B, int Ds, int Hs, int Ws, int Cu,
                                   int Dk, int Hk, int Wk, int Ck, dram_stream_t stream) {
  return upcat_bwd_impl<bf16_t>((const bf16_t*)dcat, (bf16_t*)dsrc, (bf16_t*)dskip, B, Ds, Hs, Ws, Cu, Dk, Hk, Wk, Ck,
                                stream);
}

extern "C" int dram_upproject_nblk(long long vps) {
  long long b = (vps + 1023) / 1024;
  return (int)(b > 1024 ? 1024 : (b < 1 ? 1 : b));
}

extern "C" int dram_upproject(const float* dense, const float* ess, float* out, float* partial, int B, int D, int H,
                              int W, int Do, int Ho, int Wo, dram_stream_t stream) {
  if (!dense || !ess || !out || !partial || B < 1 || D < 1 || H < 1 || W < 1 || Do < 1 || Ho < 1 || Wo < 1)
    return DRAM_ERR_BAD_ARG;
  const long vps = (long)Do * Ho * Wo;
  const int nblk = dram_upproject_nblk(vps);
  DramProf prof(DRAM_FAM_POOL_UP, 5, 0.0, 4.0 * ((double)B * D * H * W + 2.0 * B * (double)vps), (hipStream_t)stream);
  hipLaunchKernelGGL(upproject_kernel, dim3(nblk, B), dim3(256), 0, (hipStream_t)stream, dense, ess, out, partial, D, H,
                     W, Do, Ho, Wo, ac_scale(D, Do), ac_scale(H, Ho), ac_scale(W, Wo), vps, nblk);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
