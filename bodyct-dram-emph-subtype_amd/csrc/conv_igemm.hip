// conv_igemm.hip -- 3x3x3 / 1x1x1 Conv3d forward and data-gradient as an implicit GEMM
// on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32), NDHWC activations.
//
// Replaces the cuDNN/ATen kernels behind nn.Conv3d at reference med3d.py:91-100, :152-157,
// :67/:76, :226 and autograd's convolution_backward (input gradient) for the same sites.
//
// GEMM view:  M = output voxels, N = output channels, K = taps x input channels.
//   A[m][k]  = in[src(m, tap)][ci]      gathered rows (K-contiguous, zero outside the volume)
//   B[n][k]  = wp[tap][n][ci]           packed weights (K-contiguous)
// Tiling: workgroup = 256 threads (4 waves), tile 256(M) x BN(N), K-step 32 (one 128-B
// line of every gathered voxel row).  The M tile is a 4x8x8 block of the output
// *lattice with step = dilation*, so a dilated conv sees a dense 6x10x10 halo.
// Each wave owns 64 rows x BN columns = 2 x (BN/32) accumulators of 32x32.
// K order is (ci-chunk outer, tap inner): the 27 taps of one chunk re-read the same
// ~77 KB halo, which stays in the XCD's L2.
// Pipeline: register-prefetch of tile it+1 is issued before the MFMAs of tile it;
// single LDS buffer, two barriers per step; 2-3 workgroups per CU overlap each other.
//
// MFMA operand trick: within a group of 8 consecutive k, lane half h = lane>>5 reads
// k = 8g+4h .. 8g+4h+3 with ONE ds_read_b128 for A and for B; MFMA step e then
// multiplies A[.][8g+4h+e] x B[8g+4h+e][.] for both halves (k order inside the
// reduction is free as long as A and B agree).  LDS rows are padded to 36 floats:
// conflict-free for the ds_read_b128 lane groups.
#include <stdio.h>
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int BM = 256;
constexpr int BK = 32;
constexpr int LDK = BK + 4;  // padded LDS row (floats)
constexpr int TZ = 4, TY = 8, TX = 8;

__device__ __forceinline__ float4 mask4(const float4 t, const bool v) {
  const unsigned m = v ? 0xffffffffu : 0u;
  float4 r;
  r.x = __uint_as_float(__float_as_uint(t.x) & m);
  r.y = __uint_as_float(__float_as_uint(t.y) & m);
  r.z = __uint_as_float(__float_as_uint(t.z) & m);
  r.w = __uint_as_float(__float_as_uint(t.w) & m);
  return r;
}

struct IGemmGeom {
  int B, Do, Ho, Wo, No;  // tensor written (M rows x N cols)
  int Di, Hi, Wi, Ci;     // tensor gathered
  int kd, kh, kw, taps;
  int lat;                // lattice step of the M tile
  int mul, off, step;     // MODE 0/1: src = o*mul + off + t*step
  int stride, pad, dil;   // MODE 2:   src = (o + pad - t*dil)/stride when divisible
  int nz, ny, nx;         // tiles per sub-lattice axis
  int tiles_per_b;        // lat^3*nz*ny*nx
  int n_tiles;            // N tiles
  int nblk;
  int ncls;               // MODE 2: classes listed in cls_nib (0: classes in index order)
  unsigned cls_nib;       // MODE 2, lat = 2: lattice classes (rz*lat + ry)*lat + rx, most taps first, 4 bits each
};

// MODE 0: forward (any stride/dilation) and MODE 1: data-gradient with stride 1 share
// the affine source map; MODE 2: data-gradient with stride > 1 (divisibility test).
// A-tile prefetch form (measured on one MI355X, fwd 64->64 @64x128x128 / 512->512 d4):
// `valid ? *p : 0` makes hipcc emit a flat_load from a selected address; FLAT ops also count on
// lgkmcnt, so the prefetch completes before the MFMA phase starts: 112 / 128 TFLOP/s.  The two
// "cleaner" forms that keep the loads in flight during the MFMAs (unconditional global_load +
// bit-mask at store time: 109 / 117; the same issued after the first k-group: 111 / 124) were
// slower: memory returns landing in VGPRs during the fp32 MFMA phase cost more than the
// exposed latency, which the other 2 workgroups on the CU cover.
template <int BN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, float* __restrict__ stats, const float* __restrict__ add,
    const float* __restrict__ gate, const IGemmGeom g) {
  constexpr int NJ = BN / 32;
  constexpr int BQ = BN / 32;  // B-tile row passes per thread
  __shared__ __attribute__((aligned(16))) float lds[(BM + BN) * LDK];
  __shared__ int taplist[28];  // MODE 2: taps whose stride-divisibility test passes for this tile
  float* As = lds;
  float* Bs = lds + BM * LDK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;

  // ---- decode the tile --------------------------------------------------------
  int n_tile, mt, b, txi, tyi, tzi, rx, ry, rz;
  if (MODE != 2) {
    int L = xcd_remap(blockIdx.x, g.nblk);
    n_tile = L % g.n_tiles;
    mt = L / g.n_tiles;
    b = mt / g.tiles_per_b;
    int r = mt - b * g.tiles_per_b;
    txi = r % g.nx; r /= g.nx;
    tyi = r % g.ny; r /= g.ny;
    tzi = r % g.nz; r /= g.nz;
    rx = r % g.lat; r /= g.lat;
    ry = r % g.lat;
    rz = r / g.lat;
  } else {
    // A lattice class owns between 1 and (taps per axis)^3 of the taps (stride 2, k = 3: 1, 2, 4 or 8 of the 27), so its
    // workgroups differ 8 x in length.  With the class as a slow index of the XCD-remapped tile number one XCD got
    // the two lightest classes and another the two heaviest (12 tap units against 3 on one launch of 1 024
    // workgroups).  Here the class is the SLOWEST index of the dispatch order, heaviest class first: the round-robin
    // dispatch spreads every class over the eight XCDs and the short workgroups fill in behind the long ones.
    const int ncl = g.lat * g.lat * g.lat;
    const int per = g.nblk / ncl;
    const int rank = blockIdx.x / per;
    int w = xcd_remap(blockIdx.x - rank * per, per);
    const int c = g.ncls ? (int)((g.cls_nib >> (4 * rank)) & 15u) : rank;
    rx = c % g.lat;
    ry = (c / g.lat) % g.lat;
    rz = c / (g.lat * g.lat);
    n_tile = w % g.n_tiles; w /= g.n_tiles;
    txi = w % g.nx; w /= g.nx;
    tyi = w % g.ny; w /= g.ny;
    tzi = w % g.nz;
    b = w / g.nz;
    mt = b * g.tiles_per_b + ((c * g.nz + tzi) * g.ny + tyi) * g.nx + txi;
  }
  const int n0 = n_tile * BN;

  // ---- per-thread gather rows ---------------------------------------------------
  const int col4 = tid & 7;
  const int row0 = tid >> 3;  // rows row0 + 32p
  int rbase[8];               // MODE 0/1: element offset of (b, c0z, c0y, c0x, 0)
  int rmask[8];               // MODE 0/1: 9 validity bits; MODE 2: packed coords
  int rcy[(MODE == 2) ? 8 : 1], rcx[(MODE == 2) ? 8 : 1];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int row = row0 + 32 * p;
    const int zo = ((tzi * TZ + (row >> 6)) * g.lat + rz);
    const int yo = ((tyi * TY + ((row >> 3) & 7)) * g.lat + ry);
    const int xo = ((txi * TX + (row & 7)) * g.lat + rx);
    const bool rv = (zo < g.Do) & (yo < g.Ho) & (xo < g.Wo);
    if (MODE != 2) {
      const int cz = zo * g.mul + g.off, cy = yo * g.mul + g.off, cx = xo * g.mul + g.off;
      rbase[p] = (((b * g.Di + cz) * g.Hi + cy) * g.Wi + cx) * g.Ci;
      int m = 0;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const int z = cz + t * g.step, y = cy + t * g.step, x = cx + t * g.step;
        m |= ((z >= 0) & (z < g.Di)) ? (1 << t) : 0;
        m |= ((y >= 0) & (y < g.Hi)) ? (8 << t) : 0;
        m |= ((x >= 0) & (x < g.Wi)) ? (64 << t) : 0;
      }
      rmask[p] = rv ? m : 0;
    } else {
      // (o + pad) = stride*q + r with the remainder r common to the whole lattice tile: the rows keep q and the element
      // offset of (b, q), a listed tap adds the tile-uniform u = (r - t*dil)/stride (exact by the divisibility test) --
      // no per-element division, 32-bit offsets (desc_ok: < 2^31 elements; unsigned, since q may lie one step outside)
      const int qz = (zo + g.pad) / g.stride, qy = (yo + g.pad) / g.stride, qx = (xo + g.pad) / g.stride;
      rbase[p] = (int)(((((unsigned)b * g.Di + qz) * g.Hi + qy) * g.Wi + qx) * g.Ci);
      rmask[p] = rv ? qz : (1 << 28);   // a row outside the tensor fails every range test
      rcy[p] = qy;
      rcx[p] = qx;
    }
  }

  const int nchunk = g.Ci / BK;
  int ntv = g.taps;
  if (MODE == 2) {
    // the M tile is a lattice of step `stride`: (o + pad - t*dil) % stride is the same for every
    // row, so invalid taps are skipped for the whole workgroup (27 -> ~27/stride^3 taps)
    if (tid == 0) {
      int n = 0;
      for (int tz = 0; tz < g.kd; ++tz)
        for (int ty = 0; ty < g.kh; ++ty)
          for (int tx = 0; tx < g.kw; ++tx) {
            const int az = rz + g.pad - tz * g.dil, ay = ry + g.pad - ty * g.dil, ax = rx + g.pad - tx * g.dil;
            const bool ok = (((az % g.stride) + g.stride) % g.stride == 0) &
                            (((ay % g.stride) + g.stride) % g.stride == 0) &
                            (((ax % g.stride) + g.stride) % g.stride == 0);
            if (ok) taplist[n++] = (tz * g.kh + ty) * g.kw + tx;
          }
      taplist[27] = n;
    }
    __syncthreads();
    ntv = taplist[27];
  }
  const int niter = nchunk * ntv;

  // Staging registers are individually named on purpose: as arrays, hipcc's PromoteAlloca
  // moved `rb` into LDS (+8..16 KB per workgroup and a vmcnt(0) right behind the prefetch).
  float4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7;
  float4 rb0, rb1, rb2, rb3;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  ra0 = ra1 = ra2 = ra3 = ra4 = ra5 = ra6 = ra7 = make_float4(0.f, 0.f, 0.f, 0.f);
  rb0 = rb1 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);

// Always a global_load from an in-bounds address (offset 0 when the tap falls outside the
// volume) followed by a select of the VALUE: selecting between addresses made hipcc emit
// flat_load against a scratch-resident zero, and FLAT ops also count on lgkmcnt, so the
// ds_read wait in front of the MFMAs drained the prefetch.
#define IG_LOAD_A01(P)                                                                              \
  ra##P = ((rmask[P] & vb) == vb) ? *reinterpret_cast<const float4*>(in + (long)(rbase[P] + toff)) : zero4;
#define IG_LOAD_A2(P)                                                                               \
  {                                                                                                 \
    const int sz = rmask[P] + uz, sy = rcy[P] + uy, sx = rcx[P] + ux;                               \
    const bool v = ((unsigned)sz < (unsigned)g.Di) & ((unsigned)sy < (unsigned)g.Hi) &              \
                   ((unsigned)sx < (unsigned)g.Wi);                                                  \
    const int o = (int)((unsigned)rbase[P] + (unsigned)toff);                                       \
    const float4 t_ = *reinterpret_cast<const float4*>(in + (v ? o : koff));                        \
    ra##P = mask4(t_, v);                                                                            \
  }

  auto load_tile = [&](int it) __attribute__((always_inline)) {
    const int c = it / ntv;
    const int tap = (MODE == 2) ? taplist[it - c * ntv] : it - c * ntv;
    const int tz = tap / (g.kh * g.kw);
    const int trem = tap - tz * (g.kh * g.kw);
    const int ty = trem / g.kw;
    const int tx = trem - ty * g.kw;
    const int koff = c * BK + col4 * 4;
    if (MODE != 2) {
      const int toff = (((tz * g.step) * g.Hi + ty * g.step) * g.Wi + tx * g.step) * g.Ci + koff;
      const int vb = (1 << tz) | (8 << ty) | (64 << tx);
      IG_LOAD_A01(0) IG_LOAD_A01(1) IG_LOAD_A01(2) IG_LOAD_A01(3)
      IG_LOAD_A01(4) IG_LOAD_A01(5) IG_LOAD_A01(6) IG_LOAD_A01(7)
    } else {
      const int uz = ((rz + g.pad) % g.stride - tz * g.dil) / g.stride;
      const int uy = ((ry + g.pad) % g.stride - ty * g.dil) / g.stride;
      const int ux = ((rx + g.pad) % g.stride - tx * g.dil) / g.stride;
      const int toff = ((uz * g.Hi + uy) * g.Wi + ux) * g.Ci + koff;
      IG_LOAD_A2(0) IG_LOAD_A2(1) IG_LOAD_A2(2) IG_LOAD_A2(3)
      IG_LOAD_A2(4) IG_LOAD_A2(5) IG_LOAD_A2(6) IG_LOAD_A2(7)
    }
    const float* wrow = wp + ((long)tap * g.No + n0 + row0) * g.Ci + koff;
    rb0 = *reinterpret_cast<const float4*>(wrow);
    if (BQ > 1) rb1 = *reinterpret_cast<const float4*>(wrow + (long)32 * g.Ci);
    if (BQ > 2) {
      rb2 = *reinterpret_cast<const float4*>(wrow + (long)64 * g.Ci);
      rb3 = *reinterpret_cast<const float4*>(wrow + (long)96 * g.Ci);
    }
  };

  auto store_tile = [&]() __attribute__((always_inline)) {
    float* ap = &As[row0 * LDK + col4 * 4];
    *reinterpret_cast<float4*>(ap + 0 * 32 * LDK) = ra0;
    *reinterpret_cast<float4*>(ap + 1 * 32 * LDK) = ra1;
    *reinterpret_cast<float4*>(ap + 2 * 32 * LDK) = ra2;
    *reinterpret_cast<float4*>(ap + 3 * 32 * LDK) = ra3;
    *reinterpret_cast<float4*>(ap + 4 * 32 * LDK) = ra4;
    *reinterpret_cast<float4*>(ap + 5 * 32 * LDK) = ra5;
    *reinterpret_cast<float4*>(ap + 6 * 32 * LDK) = ra6;
    *reinterpret_cast<float4*>(ap + 7 * 32 * LDK) = ra7;
    float* bp = &Bs[row0 * LDK + col4 * 4];
    *reinterpret_cast<float4*>(bp) = rb0;
    if (BQ > 1) *reinterpret_cast<float4*>(bp + 32 * LDK) = rb1;
    if (BQ > 2) {
      *reinterpret_cast<float4*>(bp + 64 * LDK) = rb2;
      *reinterpret_cast<float4*>(bp + 96 * LDK) = rb3;
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  const int li = lane & 31;
  const int lh = lane >> 5;
  const float* a_rd = &As[(wave * 64 + li) * LDK + 4 * lh];
  const float* b_rd = &Bs[li * LDK + 4 * lh];

  if (niter > 0) load_tile(0);
  for (int it = 0; it < niter; ++it) {
    __syncthreads();  // every wave has finished reading the previous tile
    store_tile();
    __syncthreads();
    if (it + 1 < niter) load_tile(it + 1);  // issued ahead of the MFMAs below
#pragma unroll
    for (int gk = 0; gk < BK / 8; ++gk) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(a_rd + gk * 8);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(a_rd + 32 * LDK + gk * 8);
      f32x4 bf[NJ];
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
        bf[nj] = *reinterpret_cast<const f32x4*>(b_rd + nj * 32 * LDK + gk * 8);
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

  // ---- epilogue -------------------------------------------------------------------
  // C/D map of the 32x32 MFMA: col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5).
  float s1[NJ], s2[NJ], bv[NJ];
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) {
    s1[nj] = 0.f;
    s2[nj] = 0.f;
    bv[nj] = bias ? bias[n0 + nj * 32 + li] : 0.f;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wave * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int zo = ((tzi * TZ + (row >> 6)) * g.lat + rz);
      const int yo = ((tyi * TY + ((row >> 3) & 7)) * g.lat + ry);
      const int xo = ((txi * TX + (row & 7)) * g.lat + rx);
      const bool rv = (zo < g.Do) & (yo < g.Ho) & (xo < g.Wo);
      const long o = ((((long)b * g.Do + zo) * g.Ho + yo) * g.Wo + xo) * g.No + n0 + li;
      if (rv) {
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          float v = acc[mi][nj][e] + bv[nj];
          if (add) {
            const float av = add[o + nj * 32];
            v += gate ? (gate[o + nj * 32] > 0.f ? av : 0.f) : av;
          }
          out[o + nj * 32] = v;
          s1[nj] += v;
          s2[nj] += v * v;
        }
      }
    }
  }
  if (stats) {
    __syncthreads();  // LDS is free again
    float* red = lds; // [4 waves][2][BN]
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      const float t1 = s1[nj] + __shfl_xor(s1[nj], 32, 64);
      const float t2 = s2[nj] + __shfl_xor(s2[nj], 32, 64);
      if (lh == 0) {
        red[(wave * 2 + 0) * BN + nj * 32 + li] = t1;
        red[(wave * 2 + 1) * BN + nj * 32 + li] = t2;
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid - which * BN;
      const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c] +
                      red[(2 * 2 + which) * BN + c] + red[(3 * 2 + which) * BN + c];
      stats[((long)mt * 2 + which) * g.No + n0 + c] = v;
    }
  }
}

// =====================================================================================
// v2: the same implicit GEMM with the tiles staged by LDS-DMA (global_load_lds_dwordx4)
// instead of registers.  No VGPR staging and no ds_write pass: the prefetch of tile it+1 is
// truly asynchronous to the MFMAs of tile it (it never touches the register file), LDS is
// double-buffered and one barrier per K-step remains.
//   * LDS rows are the raw 128-B lines (no padding: an LDS-DMA wave-instruction writes 1 KiB
//     contiguously = 8 rows); bank conflicts are removed by an XOR swizzle of the 16-B slot,
//     slot' = slot ^ ((row >> 1) & 7), applied on the per-lane SOURCE address and on the
//     ds_read_b128 address (the DMA destination stays linear).
//   * taps that fall outside the volume read from a zero line in global memory.
//   * each wave DMAs exactly the 64 A rows it later multiplies; the B tile is shared.
__device__ __attribute__((aligned(128))) float g_zero_line[32];

template <int BN, int MODE>
__global__ __launch_bounds__(256, 2) void conv_igemm2_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, float* __restrict__ stats, const float* __restrict__ add,
    const float* __restrict__ gate, const IGemmGeom g) {
  static_assert(MODE == 0 || MODE == 1, "v2 covers the affine source maps");
  constexpr int NJ = BN / 32;
  constexpr int STAGE = (BM + BN) * 32;           // floats per stage
  __shared__ __attribute__((aligned(1024))) float lds[2 * STAGE];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int L = xcd_remap(blockIdx.x, g.nblk);
  const int n_tile = L % g.n_tiles;
  int mt = L / g.n_tiles;
  const int b = mt / g.tiles_per_b;
  int r = mt - b * g.tiles_per_b;
  const int txi = r % g.nx; r /= g.nx;
  const int tyi = r % g.ny; r /= g.ny;
  const int tzi = r % g.nz; r /= g.nz;
  const int rx = r % g.lat; r /= g.lat;
  const int ry = r % g.lat;
  const int rz = r / g.lat;
  const int n0 = n_tile * BN;

  // ---- the 8 A rows this lane feeds (DMA instruction j covers rows 64*wave + 8j .. +7) ----
  const int sub = lane >> 3;           // row within the 8-row DMA piece
  const int pslot = lane & 7;          // physical 16-B slot written by this lane
  const int s_even = pslot ^ (lane >> 4);   // logical slot for even j  ((row>>1)&7 = (4j + lane>>4)&7)
  const int s_odd = s_even ^ 4;
  int rbase[8], rmask[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int row = wave * 64 + j * 8 + sub;
    const int zo = ((tzi * TZ + (row >> 6)) * g.lat + rz);
    const int yo = ((tyi * TY + ((row >> 3) & 7)) * g.lat + ry);
    const int xo = ((txi * TX + (row & 7)) * g.lat + rx);
    const bool rv = (zo < g.Do) & (yo < g.Ho) & (xo < g.Wo);
    const int cz = zo * g.mul + g.off, cy = yo * g.mul + g.off, cx = xo * g.mul + g.off;
    rbase[j] = (((b * g.Di + cz) * g.Hi + cy) * g.Wi + cx) * g.Ci + ((j & 1) ? s_odd : s_even) * 4;
    int m = 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const int z = cz + t * g.step, y = cy + t * g.step, x = cx + t * g.step;
      m |= ((z >= 0) & (z < g.Di)) ? (1 << t) : 0;
      m |= ((y >= 0) & (y < g.Hi)) ? (8 << t) : 0;
      m |= ((x >= 0) & (x < g.Wi)) ? (64 << t) : 0;
    }
    rmask[j] = rv ? m : 0;
  }
  // B rows of this lane: piece jj covers rows (BN/4)*wave + 8*jj .. +7
  constexpr int BJ = BN / 32;
  int boff[BJ];
#pragma unroll
  for (int jj = 0; jj < BJ; ++jj) {
    const int nrow = wave * (BN / 4) + jj * 8 + sub;
    const int sl = pslot ^ ((nrow >> 1) & 7);
    boff[jj] = (n0 + nrow) * g.Ci + sl * 4;
  }

  const int nchunk = g.Ci / BK;
  const int niter = nchunk * g.taps;
  const float* zline = g_zero_line + pslot * 4;

  // DMA pieces of one tile: 8 A pieces (this wave's 64 rows) + BJ B pieces; piece p is issued
  // by issue_piece so that the K-loop can drop one piece into each MFMA step's shadow.
  constexpr int NP = 8 + BJ;
  int n_toff = 0, n_vb = 0;          // tap offset / validity bits of the tile being prefetched
  const float* n_wt = wp;
  float* n_as = lds;
  float* n_bs = lds;
  auto begin_tile = [&](int it, int stage) __attribute__((always_inline)) {
    const int c = it / g.taps;
    const int tap = it - c * g.taps;
    const int tz = tap / (g.kh * g.kw);
    const int trem = tap - tz * (g.kh * g.kw);
    const int ty = trem / g.kw;
    const int tx = trem - ty * g.kw;
    n_toff = (((tz * g.step) * g.Hi + ty * g.step) * g.Wi + tx * g.step) * g.Ci + c * BK;
    n_vb = (1 << tz) | (8 << ty) | (64 << tx);
    n_as = lds + stage * STAGE + wave * 64 * 32;
    n_bs = lds + stage * STAGE + BM * 32 + wave * (BN / 4) * 32;
    n_wt = wp + (long)tap * g.No * g.Ci + c * BK;
  };
#define IG2_PIECE(P)                                                                                   \
  if ((P) < 8) {                                                                                        \
    const bool v_ = (rmask[(P) & 7] & n_vb) == n_vb;                                                    \
    const float* src_ = v_ ? in + (long)(rbase[(P) & 7] + n_toff) : zline;                              \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src_,               \
                                     (__attribute__((address_space(3))) void*)(n_as + ((P) & 7) * 8 * 32), 16, 0, 0); \
  } else if ((P) < NP) {                                                                                \
    constexpr int jj_ = ((P) - 8) < BJ ? ((P) - 8) : 0;                                                 \
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(n_wt + boff[jj_]), \
                                     (__attribute__((address_space(3))) void*)(n_bs + jj_ * 8 * 32), 16, 0, 0); \
  }
  auto issue = [&](int it, int stage) __attribute__((always_inline)) {
    begin_tile(it, stage);
    IG2_PIECE(0) IG2_PIECE(1) IG2_PIECE(2) IG2_PIECE(3) IG2_PIECE(4) IG2_PIECE(5) IG2_PIECE(6) IG2_PIECE(7)
    IG2_PIECE(8) IG2_PIECE(9) IG2_PIECE(10) IG2_PIECE(11)
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  const int li = lane & 31;
  const int lh = lane >> 5;
  const int rsw = (li >> 1) & 7;   // swizzle of the rows this lane reads (row offsets are multiples of 32)
  const int a_row = (wave * 64 + li) * 32;
  const int b_row = BM * 32 + li * 32;

  // All DMA pieces of tile it+1 are issued right behind the barrier, ahead of the MFMAs of tile
  // it.  (Dropping one piece into each MFMA step's shadow measured 3-4 % SLOWER on the same
  // device: an LDS-DMA issue among ds_reads + MFMAs costs 100-185 cycles.)
  if (niter > 0) issue(0, 0);
  for (int it = 0; it < niter; ++it) {
    __syncthreads();  // tile `it` has landed (vmcnt(0) + barrier); stage (it+1)&1 is free again
    if (it + 1 < niter) issue(it + 1, (it + 1) & 1);
    const float* st = lds + (it & 1) * STAGE;
#pragma unroll
    for (int gk = 0; gk < BK / 8; ++gk) {
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
#undef IG2_PIECE

  // ---- epilogue (identical to v1) --------------------------------------------------------
  float s1[NJ], s2[NJ], bv[NJ];
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) {
    s1[nj] = 0.f;
    s2[nj] = 0.f;
    bv[nj] = bias ? bias[n0 + nj * 32 + li] : 0.f;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = wave * 64 + mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int zo = ((tzi * TZ + (row >> 6)) * g.lat + rz);
      const int yo = ((tyi * TY + ((row >> 3) & 7)) * g.lat + ry);
      const int xo = ((txi * TX + (row & 7)) * g.lat + rx);
      const bool rv = (zo < g.Do) & (yo < g.Ho) & (xo < g.Wo);
      const long o = ((((long)b * g.Do + zo) * g.Ho + yo) * g.Wo + xo) * g.No + n0 + li;
      if (rv) {
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          float v = acc[mi][nj][e] + bv[nj];
          if (add) {
            const float av = add[o + nj * 32];
            v += gate ? (gate[o + nj * 32] > 0.f ? av : 0.f) : av;
          }
          out[o + nj * 32] = v;
          s1[nj] += v;
          s2[nj] += v * v;
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    float* red = lds;  // [4 waves][2][BN]
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      const float t1 = s1[nj] + __shfl_xor(s1[nj], 32, 64);
      const float t2 = s2[nj] + __shfl_xor(s2[nj], 32, 64);
      if (lh == 0) {
        red[(wave * 2 + 0) * BN + nj * 32 + li] = t1;
        red[(wave * 2 + 1) * BN + nj * 32 + li] = t2;
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid - which * BN;
      const float v = red[(0 * 2 + which) * BN + c] + red[(1 * 2 + which) * BN + c] +
                      red[(2 * 2 + which) * BN + c] + red[(3 * 2 + which) * BN + c];
      stats[((long)mt * 2 + which) * g.No + n0 + c] = v;
    }
  }
}

// =====================================================================================
// v3: 3-D halo resident in LDS.  For stride-1 3x3x3 convolutions (and their data gradient)
// a workgroup of 8 waves owns an 8x8x8 block of the output lattice; per 32-channel chunk it
// DMAs the 10x10x10 input halo (1000 rows x 128 B) into LDS ONCE and runs all 27 taps out
// of it (the A fragment of tap (a,b,c) is the same image read at row offset (a,b,c)); only
// the small weight tile (BN rows) is streamed per tap, double-buffered, one piece per wave.
// Vector-memory instructions per MFMA: 0.156 (v1/v2) -> ~0.025, which is what bounds the
// fp32-MFMA loop on gfx950 (tools/mfma_ablate.hip).  LDS image: raw 128-B rows, 16-B slot
// XOR-swizzled with ((x_halo >> 1) + 4*(y_halo & 1)) & 7 -- conflict-free for the
// ds_read_b128 lane groups at every tap offset (checked exhaustively); the swizzle is
// applied on the DMA source address and on the read address, the DMA destination is linear.
// Template: NJ = 32-column accumulator blocks per wave, TZ3 = tile depth (8: 8x8x8 tile, waves
// 8(M) x 1(N); 4: 4x8x8 tile -- one dilation-4 residue sub-volume of the 16x32x32 stages -- waves
// 4(M) x 2(N)), so a workgroup covers 32*NJ*WN output channels.
template <int NJ, int MODE, int TZ3>
__global__ __launch_bounds__(512, 2) void conv_igemm3_kernel(
    const float* __restrict__ in, const float* __restrict__ wp, const float* __restrict__ bias,
    float* __restrict__ out, float* __restrict__ stats, const float* __restrict__ add,
    const float* __restrict__ gate, const IGemmGeom g) {
  static_assert(MODE == 0 || MODE == 1, "v3: stride-1 forward / data gradient");
  static_assert(TZ3 == 8 || TZ3 == 4, "tile depth");
  constexpr int WN = (TZ3 == 8) ? 1 : 2;        // waves along N
  constexpr int WM = 8 / WN;                    // waves along M (= tile depth)
  constexpr int BNW = NJ * 32;                  // columns per wave
  constexpr int BN = BNW * WN;                  // columns per workgroup
  constexpr int HP = 10;                        // halo pitch in y and x (8 + 2)
  constexpr int HROWS = (TZ3 + 2) * HP * HP;    // 1000 / 600
  constexpr int NPIECE = HROWS / 8;             // DMA pieces of 8 rows
  constexpr int HQ = (NPIECE + 7) / 8;          // halo pieces per wave
  constexpr int PB = (BN / 8 + 7) / 8;          // weight pieces per wave and tap
  constexpr int HALO = HROWS * 32;              // floats
  constexpr int BST = BN * 32;                  // floats per weight stage
  __shared__ __attribute__((aligned(1024))) float lds[HALO + 2 * BST];
  float* halo = lds;
  float* bst = lds + HALO;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WM;                     // tile z-plane of this wave's 64 rows
  const int wn = wave / WM;                     // N half

  int L = xcd_remap(blockIdx.x, g.nblk);
  const int n_tile = L % g.n_tiles;
  int mt = L / g.n_tiles;
  const int b = mt / g.tiles_per_b;
  int r = mt - b * g.tiles_per_b;
  const int txi = r % g.nx; r /= g.nx;
  const int tyi = r % g.ny; r /= g.ny;
  const int tzi = r % g.nz; r /= g.nz;
  const int rx = r % g.lat; r /= g.lat;
  const int ry = r % g.lat;
  const int rz = r / g.lat;
  const int n0 = n_tile * BN;

  // ---- halo DMA pieces of this lane: p = wave + 8q -----------------------------------------
  const int sub = lane >> 3, pslot = lane & 7;
  int hoff[HQ];  // element offset of (row, logical slot) at channel chunk 0, or -1 (zero row)
#pragma unroll
  for (int q = 0; q < HQ; ++q) {
    const int row = 8 * (wave + 8 * q) + sub;
    const int zh = row / (HP * HP), yh = (row / HP) % HP, xh = row % HP;
    const int zi = (tzi * TZ3 + zh - 1) * g.lat + rz;
    const int yi = (tyi * 8 + yh - 1) * g.lat + ry;
    const int xi = (txi * 8 + xh - 1) * g.lat + rx;
    const bool v = (row < HROWS) & (tzi * TZ3 + zh >= 1) & (tyi * 8 + yh >= 1) & (txi * 8 + xh >= 1) &
                   (zi < g.Di) & (yi < g.Hi) & (xi < g.Wi);
    const int sl = pslot ^ (((xh >> 1) + 4 * (yh & 1)) & 7);
    hoff[q] = v ? (((b * g.Di + zi) * g.Hi + yi) * g.Wi + xi) * g.Ci + sl * 4 : -1;
  }
  const float* zline = g_zero_line + pslot * 4;
  // weight pieces of this wave: piece pb = wave + 8j covers rows 8*pb .. 8*pb+7 of the BN-row tile
  int boff[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    const int nrow = (wave + 8 * j) * 8 + sub;
    boff[j] = (n0 + (nrow < BN ? nrow : 0)) * g.Ci + (pslot ^ ((nrow >> 1) & 7)) * 4;
  }

  auto issue_halo = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < HQ; ++q) {
      if (wave + 8 * q < NPIECE) {  // wave-uniform
        const float* src = hoff[q] >= 0 ? in + (long)(hoff[q] + c * BK) : zline;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(halo + (wave + 8 * q) * 256), 16, 0, 0);
      }
    }
  };
  auto issue_b = [&](int c, int tap, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      if ((wave + 8 * j) * 8 < BN) {  // wave-uniform
        const float* src = wp + (long)tap * g.No * g.Ci + c * BK + boff[j];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(bst + stage * BST + (wave + 8 * j) * 256), 16, 0, 0);
      }
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nj][e] = 0.f;

  const int li = lane & 31, lh = lane >> 5;
  const int tyl0 = li >> 3, txl = li & 7;       // mi adds 4 to tyl
  const int rsw = (li >> 1) & 7;                // weight rows: swizzle of row (.. + li), offsets are multiples of 32
  const int nchunk = g.Ci / BK;

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                            // every wave is done with the previous chunk's halo
    issue_halo(c);
    issue_b(c, 0, 0);
    for (int t = 0; t < 27; ++t) {
      __syncthreads();                          // weight tile t (and, at t = 0, the halo) has landed
      if (t + 1 < 27) issue_b(c, t + 1, (t + 1) & 1);
      const int tz = t / 9, ty = (t / 3) % 3, tx = t % 3;
      const int oz = (MODE == 0) ? tz : 2 - tz, oy = (MODE == 0) ? ty : 2 - ty, ox = (MODE == 0) ? tx : 2 - tx;
      const int xh = txl + ox;
      const int yh0 = tyl0 + oy;
      const int ra0 = (((wm + oz) * HP + yh0) * HP + xh) * 32;
      const int ra1 = ra0 + 4 * HP * 32;
      const int g0 = ((xh >> 1) + 4 * (yh0 & 1)) & 7;   // rows yh0 and yh0+4: same parity, same swizzle
      const float* sb = bst + (t & 1) * BST + (wn * BNW + li) * 32;
#pragma unroll
      for (int gk = 0; gk < BK / 8; ++gk) {
        const int sa = ((2 * gk + lh) ^ g0) * 4;
        const int sw = ((2 * gk + lh) ^ rsw) * 4;
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(halo + ra0 + sa);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(halo + ra1 + sa);
        f32x4 bf[NJ];
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) bf[nj] = *reinterpret_cast<const f32x4*>(sb + nj * 32 * 32 + sw);
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
  }

  // ---- epilogue: this wave's rows are (z = wm, y = 4*mi + .., x), columns n0 + wn*BNW .. ------
  const int nb = n0 + wn * BNW;
  float s1[NJ], s2[NJ], bv[NJ];
#pragma unroll
  for (int nj = 0; nj < NJ; ++nj) {
    s1[nj] = 0.f;
    s2[nj] = 0.f;
    bv[nj] = bias ? bias[nb + nj * 32 + li] : 0.f;
  }
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = mi * 32 + (e & 3) + 8 * (e >> 2) + 4 * lh;   // within the wave's 64 rows
      const int zo = ((tzi * TZ3 + wm) * g.lat + rz);
      const int yo = ((tyi * 8 + (row >> 3)) * g.lat + ry);
      const int xo = ((txi * 8 + (row & 7)) * g.lat + rx);
      const bool rv = (zo < g.Do) & (yo < g.Ho) & (xo < g.Wo);
      const long o = ((((long)b * g.Do + zo) * g.Ho + yo) * g.Wo + xo) * g.No + nb + li;
      if (rv) {
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj) {
          float v = acc[mi][nj][e] + bv[nj];
          if (add) {
            const float av = add[o + nj * 32];
            v += gate ? (gate[o + nj * 32] > 0.f ? av : 0.f) : av;
          }
          out[o + nj * 32] = v;
          s1[nj] += v;
          s2[nj] += v * v;
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    float* red = lds;  // [8 waves][2][BNW]
#pragma unroll
    for (int nj = 0; nj < NJ; ++nj) {
      const float t1 = s1[nj] + __shfl_xor(s1[nj], 32, 64);
      const float t2 = s2[nj] + __shfl_xor(s2[nj], 32, 64);
      if (lh == 0) {
        red[(wave * 2 + 0) * BNW + nj * 32 + li] = t1;
        red[(wave * 2 + 1) * BNW + nj * 32 + li] = t2;
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, c = tid - which * BN;   // c: column within the workgroup's BN
      const int cwn = c / BNW, cc = c - cwn * BNW;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) v += red[((cwn * WM + w) * 2 + which) * BNW + cc];
      stats[((long)mt * 2 + which) * g.No + n0 + c] = v;
    }
  }
}

// Tile the dilation lattice only when a sub-lattice fills a 4x8x8 tile; on small volumes
// (e.g. dilation 4 on 8x16x16) lattice tiles would be mostly empty, plain tiles are denser.
int choose_lat(int dil, int Do, int Ho, int Wo) {
  if (dil <= 1) return 1;
  const int sz = (Do + dil - 1) / dil, sy = (Ho + dil - 1) / dil, sx = (Wo + dil - 1) / dil;
  return (sz >= TZ && sy >= TY && sx >= TX) ? dil : 1;
}

int pick_bn(int N) {
  if (N % 128 == 0) return 128;
  if (N % 64 == 0) return 64;
  if (N % 32 == 0) return 32;
  return 0;
}

// Narrow the N tile while the launch would not fill the chip (256 CUs x 2-3 workgroups):
// the 16x32x32 stages have only 128 M tiles at batch 2.
int igemm_version() {   // DRAM_IGEMM_V: 0 auto, 1 register-staged, 2 LDS-DMA tiles, 3 LDS-resident halo
  static int ver = -1;
  if (ver < 0) {
    const char* v = tune_env("DRAM_IGEMM_V");
    ver = v ? atoi(v) : 0;
  }
  return ver;
}

// v3 plan for a stride-1 3x3x3 "same" convolution writing N channels: tile depth and N per
// workgroup.  Returns false when v3 does not apply.
struct V3Plan { int tz3, bn; };
bool plan_v3(const DramConvDesc* d, int N, int Do, int Ho, int Wo, V3Plan& p) {
  const int ver = igemm_version();
  if (ver != 0 && ver != 3) return false;
  if (d->k != 3 || d->stride != 1 || d->pad != d->dil) return false;
  if (const char* f = tune_env("DRAM_IGEMM_V3_FORCE")) {   // tests / tuning: "tz3,bn", e.g. "4,256"
    int tz = 0, bn = 0;
    if (sscanf(f, "%d,%d", &tz, &bn) == 2 && (tz == 4 || tz == 8) && N % bn == 0 &&
        ((tz == 8 && (bn == 32 || bn == 64)) || (tz == 4 && (bn == 128 || bn == 256)))) {
      p = {tz, bn};
      return true;
    }
    return false;
  }
  const int lat = d->dil;
  const int sz = (Do + lat - 1) / lat, sy = (Ho + lat - 1) / lat, sx = (Wo + lat - 1) / lat;
  if (N == 32) { p = {8, 32}; return true; }
  if (N % 64 != 0) return false;
  const long sub8 = (long)d->B * lat * lat * lat * ((sz + 7) / 8) * ((sy + 7) / 8) * ((sx + 7) / 8);
  const long sub4 = (long)d->B * lat * lat * lat * ((sz + 3) / 4) * ((sy + 7) / 8) * ((sx + 7) / 8);
  // wide layers on small volumes (16x32x32 stages): 4x8x8 tiles, 128 or 256 columns per workgroup
  if (N % 256 == 0 && sub4 * (N / 256) >= 192 && sub8 * (N / 64) < 4096 && sz % 8 != 0) { p = {4, 256}; return true; }
  if (N % 128 == 0 && sub8 * (N / 64) < 1024) {
    if (sub4 * (N / 128) >= 192) { p = {4, 128}; return true; }
    return false;
  }
  if (sub8 * (N / 64) >= 256) { p = {8, 64}; return true; }
  return false;
}

int pick_bn_for(const IGemmGeom& g0) {
  int BN = pick_bn(g0.No);
  if (!BN) return 0;
  IGemmGeom g = g0;
  while (BN > 32) {
    const int sz = (g.Do + g.lat - 1) / g.lat, sy = (g.Ho + g.lat - 1) / g.lat, sx = (g.Wo + g.lat - 1) / g.lat;
    const long mt = (long)g.B * g.lat * g.lat * g.lat * ((sz + TZ - 1) / TZ) * ((sy + TY - 1) / TY) * ((sx + TX - 1) / TX);
    if (mt * (g.No / BN) >= 512) break;
    BN >>= 1;
  }
  return BN;
}

void fill_tiles(IGemmGeom& g, int BN, int tz_edge = TZ) {
  const int sz = (g.Do + g.lat - 1) / g.lat, sy = (g.Ho + g.lat - 1) / g.lat, sx = (g.Wo + g.lat - 1) / g.lat;
  g.nz = (sz + tz_edge - 1) / tz_edge;
  g.ny = (sy + TY - 1) / TY;
  g.nx = (sx + TX - 1) / TX;
  g.tiles_per_b = g.lat * g.lat * g.lat * g.nz * g.ny * g.nx;
  g.n_tiles = g.No / BN;
  g.nblk = g.B * g.tiles_per_b * g.n_tiles;
}

bool desc_ok(const DramConvDesc* d) {
  if (!d) return false;
  if (d->B < 1 || d->D < 1 || d->H < 1 || d->W < 1 || d->Cin < 1 || d->Cout < 1) return false;
  if (d->k != 1 && d->k != 3) return false;
  if (d->stride < 1 || d->dil < 1 || d->pad < 0) return false;
  const int eff = d->dil * (d->k - 1) + 1;
  auto od = [&](int n) { return (n + 2 * d->pad - eff) / d->stride + 1; };
  if (d->Do != od(d->D) || d->Ho != od(d->H) || d->Wo != od(d->W)) return false;
  // int32 element offsets inside the kernels
  const long long ein = (long long)d->B * d->D * d->H * d->W * d->Cin;
  const long long eout = (long long)d->B * d->Do * d->Ho * d->Wo * d->Cout;
  if (ein >= (1LL << 31) || eout >= (1LL << 31)) return false;
  return true;
}

// Workgroups per CU the register/LDS budget admits (see -Rpass-analysis): BN=128 -> 2, else 3.
// When the grid is a poor multiple of 256 CUs x that many slots (e.g. 1024 tiles on 768 slots
// = 1.33 rounds) but a good multiple of 2 per CU, pad the launch with dynamic LDS so that only
// two workgroups fit per CU: 1024 tiles then run as exactly two full rounds.
int lds_pad_for_balance(int BN, int nblk) {
  if (BN == 128) return 0;
  auto eff = [&](int slots) { return (double)nblk / (double)(((nblk + slots - 1) / slots) * slots); };
  const double e3 = eff(768), e2 = eff(512) * 0.97;  // 2/CU overlaps slightly less
  if (e2 <= e3) return 0;
  const int stat = (BM + BN) * LDK * 4;
  const int need = 160 * 1024 / 3 + 1024;  // > 1/3 of the 160 KB LDS
  return need > stat ? need - stat : 0;
}

// timeline tags: executed = algorithmic 2*M*N*K (the strided data gradient touches 1/stride^3 of the taps per
// voxel); bytes: gathered tensor + written tensor (+ add, gate) + packed weights, once each
inline double igemm_flops(const IGemmGeom& g, int mode) {
  const double st3 = mode == 2 ? (double)g.stride * g.stride * g.stride : 1.0;
  return 2.0 * g.B * g.Do * g.Ho * g.Wo * (double)g.No * g.Ci * g.taps / st3;
}
inline double igemm_bytes(const IGemmGeom& g, const float* add, const float* gate) {
  return 4.0 * ((double)g.B * g.Di * g.Hi * g.Wi * g.Ci +
                (double)g.B * g.Do * g.Ho * g.Wo * g.No * (1.0 + (add ? 1 : 0) + (gate ? 1 : 0)) +
                (double)g.taps * g.No * g.Ci);
}

template <int MODE>
int launch(int BN, const float* in, const float* wp, const float* bias, float* out, float* stats,
           const float* add, const float* gate, IGemmGeom& g, hipStream_t s) {
  fill_tiles(g, BN);
  DramProf prof(DRAM_FAM_CONV_IGEMM, 1000 * MODE + BN, igemm_flops(g, MODE), igemm_bytes(g, add, gate), s);
  dim3 grid(g.nblk), block(256);
  const int pad = lds_pad_for_balance(BN, g.nblk);
  const int ver = igemm_version();   // 0 = auto: LDS-DMA kernel for BN <= 64, register-staged for BN = 128
  if ((ver == 2 || ((ver == 0 || ver == 3) && BN <= 64)) && MODE != 2) {
    constexpr int M2 = MODE == 2 ? 0 : MODE;
    switch (BN) {
      case 128:
        hipLaunchKernelGGL((conv_igemm2_kernel<128, M2>), grid, block, 0, s, in, wp, bias, out, stats, add, gate, g);
        break;
      case 64:
        hipLaunchKernelGGL((conv_igemm2_kernel<64, M2>), grid, block, 0, s, in, wp, bias, out, stats, add, gate, g);
        break;
      case 32:
        hipLaunchKernelGGL((conv_igemm2_kernel<32, M2>), grid, block, 0, s, in, wp, bias, out, stats, add, gate, g);
        break;
      default:
        return DRAM_ERR_UNSUPPORTED;
    }
    DRAM_LAUNCH_CHECK();
    return DRAM_OK;
  }
#define IG_LAUNCH(BN_) \
  hipLaunchKernelGGL((conv_igemm_kernel<BN_, MODE>), grid, block, pad, s, in, wp, bias, out, stats, add, gate, g)
  switch (BN) {
    case 128: IG_LAUNCH(128); break;
    case 64: IG_LAUNCH(64); break;
    case 32: IG_LAUNCH(32); break;
    default: return DRAM_ERR_UNSUPPORTED;
  }
#undef IG_LAUNCH
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

template <int MODE>
int launch3(const V3Plan& p, const float* in, const float* wp, const float* bias, float* out, float* stats,
            const float* add, const float* gate, IGemmGeom& g, hipStream_t s) {
  fill_tiles(g, p.bn, p.tz3);
  dim3 grid(g.nblk), block(512);
  DramProf prof(DRAM_FAM_CONV_IGEMM, 3000 + 1000 * MODE + p.bn, igemm_flops(g, MODE), igemm_bytes(g, add, gate), s);
#define IG3(NJ_, TZ_) \
  hipLaunchKernelGGL((conv_igemm3_kernel<NJ_, MODE, TZ_>), grid, block, 0, s, in, wp, bias, out, stats, add, gate, g)
  if (p.tz3 == 8 && p.bn == 64) IG3(2, 8);
  else if (p.tz3 == 8 && p.bn == 32) IG3(1, 8);
  else if (p.tz3 == 4 && p.bn == 128) IG3(2, 4);
  else if (p.tz3 == 4 && p.bn == 256) IG3(4, 4);
  else return DRAM_ERR_UNSUPPORTED;
#undef IG3
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

}  // namespace

extern "C" int dram_conv_num_mtiles(const DramConvDesc* d) {
  if (!desc_ok(d)) return DRAM_ERR_BAD_ARG;
  IGemmGeom g{};
  g.B = d->B; g.Do = d->Do; g.Ho = d->Ho; g.Wo = d->Wo; g.No = d->Cout;
  g.lat = d->dil;
  const int BN = pick_bn(d->Cout);
  if (!BN) return DRAM_ERR_UNSUPPORTED;
  // the M-tile count does not depend on the N tile, but it does on the kernel: v3 owns 8x8x8 / 4x8x8 tiles
  V3Plan p3;
  if (plan_v3(d, d->Cout, d->Do, d->Ho, d->Wo, p3)) {
    fill_tiles(g, p3.bn, p3.tz3);
    return g.B * g.tiles_per_b;
  }
  g.lat = choose_lat(d->dil, d->Do, d->Ho, d->Wo);
  fill_tiles(g, BN);
  return g.B * g.tiles_per_b;
}

extern "C" int dram_conv3d_fwd(const float* x, const float* wf, const float* bias, float* y,
                               float* stats_partial, const DramConvDesc* d, dram_stream_t stream) {
  if (!desc_ok(d) || !x || !wf || !y) return DRAM_ERR_BAD_ARG;
  if (d->Cin % BK != 0) return DRAM_ERR_UNSUPPORTED;
  IGemmGeom g{};
  g.B = d->B; g.Do = d->Do; g.Ho = d->Ho; g.Wo = d->Wo; g.No = d->Cout;
  g.Di = d->D; g.Hi = d->H; g.Wi = d->W; g.Ci = d->Cin;
  g.kd = g.kh = g.kw = d->k; g.taps = d->k * d->k * d->k;
  g.lat = d->dil;
  g.mul = d->stride; g.off = -d->pad; g.step = d->dil;
  g.stride = d->stride; g.pad = d->pad; g.dil = d->dil;
  V3Plan p3;
  if (plan_v3(d, d->Cout, d->Do, d->Ho, d->Wo, p3))
    return launch3<0>(p3, x, wf, bias, y, stats_partial, nullptr, nullptr, g, (hipStream_t)stream);
  g.lat = choose_lat(d->dil, d->Do, d->Ho, d->Wo);
  const int BN = pick_bn_for(g);
  if (!BN) return DRAM_ERR_UNSUPPORTED;
  return launch<0>(BN, x, wf, bias, y, stats_partial, nullptr, nullptr, g, (hipStream_t)stream);
}

extern "C" int dram_conv3d_bwd_data(const float* dy, const float* wb, float* dx, const float* add,
                                    const float* gate, const DramConvDesc* d, dram_stream_t stream) {
  if (!desc_ok(d) || !dy || !wb || !dx) return DRAM_ERR_BAD_ARG;
  if (gate && !add) return DRAM_ERR_BAD_ARG;
  if (d->Cout % BK != 0) return DRAM_ERR_UNSUPPORTED;
  if (!pick_bn(d->Cin)) return DRAM_ERR_UNSUPPORTED;
  IGemmGeom g{};
  // the tensor written is dx (forward input grid); the tensor gathered is dy
  g.B = d->B; g.Do = d->D; g.Ho = d->H; g.Wo = d->W; g.No = d->Cin;
  g.Di = d->Do; g.Hi = d->Ho; g.Wi = d->Wo; g.Ci = d->Cout;
  g.kd = g.kh = g.kw = d->k; g.taps = d->k * d->k * d->k;
  g.stride = d->stride; g.pad = d->pad; g.dil = d->dil;
  if (d->stride == 1) {
    g.lat = d->dil;
    g.mul = 1; g.off = d->pad; g.step = -d->dil;
    V3Plan p3;
    if (plan_v3(d, d->Cin, d->D, d->H, d->W, p3))
      return launch3<1>(p3, dy, wb, nullptr, dx, nullptr, add, gate, g, (hipStream_t)stream);
    g.lat = choose_lat(d->dil, d->D, d->H, d->W);
    return launch<1>(pick_bn_for(g), dy, wb, nullptr, dx, nullptr, add, gate, g, (hipStream_t)stream);
  }
  g.lat = d->stride;
  {  // lattice classes by tap count, most first (ties in index order); listed for stride 2, index order otherwise
    const int L = g.lat, n = L * L * L;
    g.ncls = 0;
    g.cls_nib = 0;
    if (n <= 8) {
      int wgt[8], ord[8];
      auto axis = [&](int r) {
        int c = 0;
        for (int t = 0; t < d->k; ++t) c += ((r + d->pad - t * d->dil) % L + L) % L == 0;
        return c;
      };
      for (int c = 0; c < n; ++c) {
        wgt[c] = axis(c / (L * L)) * axis((c / L) % L) * axis(c % L);
        ord[c] = c;
      }
      for (int i = 1; i < n; ++i)   // insertion sort: stable
        for (int j = i; j > 0 && wgt[ord[j]] > wgt[ord[j - 1]]; --j) {
          const int t = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = t;
        }
      for (int c = 0; c < n; ++c) g.cls_nib |= (unsigned)ord[c] << (4 * c);
      g.ncls = n;
    }
  }
  return launch<2>(pick_bn_for(g), dy, wb, nullptr, dx, nullptr, add, gate, g, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------
// weight repack: [Cout][Cin][taps] -> wf[tap][Cout][Cin], wb[tap][Cin][Cout]
namespace {
__global__ void pack_weight_kernel(const float* __restrict__ w, float* __restrict__ wf,
                                   float* __restrict__ wb, int Cout, int Cin, int taps) {
  const long n = (long)Cout * Cin * taps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // i enumerates wf order: (tap, co, ci) -> coalesced writes of wf, strided reads of w
    const int ci = (int)(i % Cin);
    const long r = i / Cin;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout);
    const float v = w[((long)co * Cin + ci) * taps + tap];
    if (wf) wf[i] = v;
    if (wb) wb[((long)tap * Cin + ci) * Cout + co] = v;
  }
}
}  // namespace

extern "C" int dram_pack_conv_weight(const float* w, float* wf, float* wb, int Cout, int Cin, int taps,
                                     dram_stream_t stream) {
  if (!w || (!wf && !wb) || Cout < 1 || Cin < 1 || taps < 1) return DRAM_ERR_BAD_ARG;
  const long n = (long)Cout * Cin * taps;
  const int grid = (int)((n + 255) / 256 > 4096 ? 4096 : (n + 255) / 256);
  DramProf prof(DRAM_FAM_WEIGHT_PACK, 0, 0.0, 4.0 * (double)n * (1.0 + (wf ? 1 : 0) + (wb ? 1 : 0)), (hipStream_t)stream);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, wf, wb, Cout, Cin, taps);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
