// conv_wino2d.hip -- fused in-plane Winograd F(2x2, 3x3) x direct-z convolution for the narrow
// (<= 128-channel) stride-1 3x3x3 layers on large volumes: forward and data gradient in ONE
// kernel, no Winograd-domain tensor in HBM.
//
// Same call sites as conv_igemm.hip (reference med3d.py:91-100 conv3x3x3 in layer1 and
// med3d.py:67/:76 decoder convs; autograd's convolution_backward (input)).  The non-fused 3-D
// Winograd path (conv_wino.hip) moves 8x the activation bytes through HBM, which only pays for
// >= 128-channel layers; here the transforms live in registers/LDS:
//   * workgroup = 8 waves, output tile 16(z) x 8 x 8 voxels x BN channels; per 16-channel chunk the
//     18x10x10 input halo (1800 rows x 64 B) is DMA'd into LDS once (as in conv_igemm3_kernel);
//   * a wave owns 2 z-planes x (4x4) in-plane 2x2 tiles = 32 MFMA rows.  For each of the 16
//     in-plane Winograd points xi = (xi_y, xi_x) and each z-tap a, the A fragment
//     (B^T v B)[xi] = +-v[p0][q0] +- v[p0][q1] +- v[p1][q0] +- v[p1][q1] is formed on the fly from
//     four ds_read_b128 of the raw halo (every row of B^T has two non-zeros);
//   * the products of one xi (3 z-taps x 16 channels) are summed in a scratch accumulator that
//     starts from 0, then added with the +-1 coefficients of A^T into the four OUTPUT accumulators
//     (y, x parity) -- so a wave keeps 4 output + 1 scratch accumulators per 32 columns instead of
//     16, and the output transform costs VALU adds in the shadow of the MFMAs;
//   * transformed weights U2[xi][a][n][k] = (G w_a G^T)[xi] are streamed per xi, double-buffered.
// MFMA work: 16 x 3 = 48 instead of 4 x 27 = 108 products per 2x2 tile -> 2.25x fewer.
// LDS images are raw 64-B rows; bank conflicts are removed by placing voxel row r at
// r ^ ((y_halo >> 1) & 1) and XOR-ing the 16-B slot with ((x_halo >> 2) & 1) | 2 * ((y_halo >> 2) & 1)
// (so the 4 x 4 in-plane tiles of a 16-lane read group land on 16 different 16-B columns; weights:
// bswz(n)), applied on the DMA source side and on the reads (SQ_LDS_BANK_CONFLICT = 0, measured).
// Issuing the next group's LDS reads ahead of the current group's MFMAs (software pipelining inside
// a wave) measured 2-4 % SLOWER; the sibling wave on the SIMD covers the latency.
// The data gradient is the same kernel on dy with the tap-flipped, transposed weights.
#include <stdlib.h>
#include "common.h"

namespace {

constexpr int W2_TZ16 = 16;                             // tile depth of the 8-wave kernel (18 x 10 x 10 halo voxels,
                                                        // DMA pieces of 16 rows x 64 B); 8 for the other variants
constexpr int W2_BK = 16;                               // channels per chunk

struct W2dGeom {
  int B, D, H, W;      // voxel grid (input == output grid)
  int Ci, No;          // gathered channels (K), written channels (N)
  int nz, ny, nx, tiles_per_b, n_tiles, nblk;
};

__device__ __attribute__((aligned(16))) float g_w2d_zero[4];

// B^T rows: two non-zeros each -> (position, sign) pairs
__device__ constexpr int kBP[4][2] = {{0, 2}, {1, 2}, {1, 2}, {1, 3}};
__device__ constexpr float kBS[4][2] = {{1.f, -1.f}, {1.f, 1.f}, {-1.f, 1.f}, {1.f, -1.f}};
// A^T = [1 1 1 0; 0 1 -1 -1]
__device__ constexpr float kAT[2][4] = {{1.f, 1.f, 1.f, 0.f}, {0.f, 1.f, -1.f, -1.f}};

// 16-B slot swizzle of weight row n.  A ds_read_b128 is served in four groups of 16 lanes
// ({0-3,12-15,20-27}, {4-11,16-19,28-31}, +32), one 256-B LDS cycle each when the 16 lanes hit 16
// different 16-B columns; with 64-B rows the column is (row & 3) * 4 + slot, so the four 4-lane runs
// of a group (rows >> 2 = {0,3,5,6} or {1,2,4,7}) need four different slot swizzles.
__device__ __forceinline__ int bswz(int n) {
  const int v = (n >> 2) & 7;
  return (v & 3) ^ ((v >> 2) * 3);
}

// y += t / y -= t on 16 floats as 8 v_pk_add_f32 (the compiler emits 16 scalar v_sub_f32 for the
// subtraction).  Inline asm is invisible to the MFMA hazard recognizer: the caller places the
// wait states (mfma_result_wait) between the last MFMA writing t and the first of these.
template <bool NEG>
__device__ __forceinline__ void pk_acc(f32x16& y, const f32x16& t) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    f32x2 yy = {y[2 * i], y[2 * i + 1]};
    const f32x2 tt = {t[2 * i], t[2 * i + 1]};
    if (NEG) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(yy) : "v"(tt));
    else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(yy) : "v"(tt));
    y[2 * i] = yy[0];
    y[2 * i + 1] = yy[1];
  }
}
// v_mfma_f32_32x32x2_f32 (16 passes) result -> VALU read: 18 wait states.  The scratch accumulators are
// operands of the asm, so it stays between the MFMAs that write them and the adds that read them.
template <int NJ>
__device__ __forceinline__ void mfma_result_wait(f32x16 (&t)[NJ]) {
  if (NJ == 2) asm volatile("s_nop 15\n\ts_nop 3" : "+v"(t[0]), "+v"(t[NJ - 1]));
  else asm volatile("s_nop 15\n\ts_nop 3" : "+v"(t[0]));
}

// address of the second k-group: slot ^ 2, i.e. float index ^ 8 -- as an opaque VALU op at the use
// site, so that the compiler does not keep a second copy of every address in registers
__device__ __forceinline__ int kx(int off, int gk) {
  if (gk == 0) return off;
  int r;
  asm volatile("v_xor_b32 %0, 8, %1" : "=v"(r) : "v"(off));
  return r;
}

// NW waves per workgroup, tile depth 2 * NW; WS weight stages.  <NJ, 8, 2> is the kernel described above (one
// 147-KB workgroup per CU).  <NJ, 4, 1>: 8-deep tiles, 256 threads, 63-KB halo + ONE weight stage = 79 KB, so
// TWO workgroups share a CU and run out of phase: the barrier / weight-DMA / halo-reload stretches of one
// fall into the MFMA stretches of the other (the two waves of a SIMD belong to different workgroups).
template <int NJ, int NW = 8, int WS = 2>   // WS: 2 = two whole-point stages, 1 = one, 3 = one stage refilled tap by tap
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv_wino2d_kernel(const float* __restrict__ in, const float* __restrict__ u2,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              float* __restrict__ stats, const float* __restrict__ add,
                                                              const float* __restrict__ gate, const W2dGeom g) {
  constexpr int W2_TZ = 2 * NW;
  constexpr int W2_HROWS = (W2_TZ + 2) * 10 * 10;
  constexpr int W2_NPIECE = (W2_HROWS + 15) / 16;
  constexpr int W2_HQ = (W2_NPIECE + NW - 1) / NW;
  constexpr int W2_HALO = W2_NPIECE * 256;
  constexpr int BN = 32 * NJ;
  constexpr int BROWS = 3 * BN;             // weight rows per xi: (z-tap, n)
  constexpr int BST = BROWS * 16;           // floats per weight stage
  constexpr int BPIECE = BROWS / 16;        // 6 * NJ DMA pieces
  constexpr int PB = (BPIECE + NW - 1) / NW;
  static_assert(WS != 3 || (NW == 4 && NJ == 2), "tap ring: one DMA piece per wave and tap");
  // SEPARATE LDS arrays for the halo and each weight stage (tap slot), not one array carved by offsets: the
  // wait-count pass asks alias analysis whether a ds_read may touch the target of an outstanding LDS-DMA,
  // and only distinct objects answer no.  Carved from one array, every point began with "s_waitcnt vmcnt(0)"
  // right behind the DMA issue of the NEXT point's weights -- the double buffering was waited away (the
  // 1 460 "barrier wait" cycles per point of the phase timing in DESIGN.md).
  constexpr int TAP = BN * 16;               // floats per z-tap slot
  __shared__ __attribute__((aligned(1024))) float lds[W2_HALO];
  __shared__ __attribute__((aligned(1024))) float wsa0[TAP], wsa1[TAP], wsa2[TAP];       // stage 0, taps 0..2
  __shared__ __attribute__((aligned(1024))) float wsb0[WS == 2 ? TAP : 64], wsb1[WS == 2 ? TAP : 64],
      wsb2[WS == 2 ? TAP : 64];                                                       // stage 1
  __shared__ int htab[W2_NPIECE * 16];
  float* halo = lds;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int L = xcd_remap(blockIdx.x, g.nblk);
  const int n_tile = L % g.n_tiles;
  const int mt = L / g.n_tiles;
  const int b = mt / g.tiles_per_b;
  int r = mt - b * g.tiles_per_b;
  const int txi = r % g.nx; r /= g.nx;
  const int tyi = r % g.ny;
  const int tzi = r / g.ny;
  const int n0 = n_tile * BN;
  const int z0 = tzi * W2_TZ, y0 = tyi * 8, x0 = txi * 8;

  // ---- halo DMA: piece p = wave + 8q, this lane feeds LDS row 16p + (lane >> 2), slot lane & 3.
  // The (row -> source voxel) map is built once per workgroup into a small LDS table: element
  // offset of the voxel at channel 0 with the slot swizzle in its two low bits
  // (offsets are multiples of Ci >= 16), or -1 for rows outside the volume.
  const int pslot = lane & 3;
  for (int rp = tid; rp < W2_NPIECE * 16; rp += 64 * NW) {
    const int yh = (rp / 10) % 10;
    const int rl = rp ^ ((yh >> 1) & 1);                      // logical row = (zh*10 + yh)*10 + xh
    const int zh = rl / 100, xh = rl % 10;
    const int zi = z0 + zh - 1, yi = y0 + yh - 1, xi = x0 + xh - 1;
    const bool v = (rp < W2_HROWS) & (zi >= 0) & (zi < g.D) & (yi >= 0) & (yi < g.H) & (xi >= 0) & (xi < g.W);
    htab[rp] = v ? ((((b * g.D + zi) * g.H + yi) * g.W + xi) * g.Ci) | ((xh >> 2) & 1) | (((yh >> 2) & 1) << 1) : -1;
  }
  // weight pieces: piece pb = wave + 8j covers rows 16*pb .. +15 of the (3 x BN)-row tile
  int boff[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    int row = 16 * (wave + NW * j) + (lane >> 2);
    if (row >= BROWS) row = 0;
    const int a = row / BN, n = row - a * BN;
    boff[j] = (a * g.No + n0 + n) * g.Ci + (pslot ^ bswz(n)) * 4;
  }
  const long xi_stride = 3L * g.No * g.Ci;

  auto issue_halo = [&](int c) __attribute__((always_inline)) {
#pragma unroll 1   // a rolled loop: the unrolled address math of 15 pieces spilled accumulators to scratch
    for (int q = 0; q < W2_HQ; ++q) {
      if (wave + NW * q < W2_NPIECE) {   // wave-uniform
        const int ho = htab[16 * (wave + NW * q) + (lane >> 2)];
        const float* src = ho >= 0 ? in + (long)((ho & ~3) + ((pslot ^ (ho & 3)) << 2) + c * W2_BK) : g_w2d_zero;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(halo + (wave + NW * q) * 256), 16, 0,
                                         0);
      }
    }
  };
  // pieces of 16 rows; a z-tap slot holds BN rows = BN / 16 pieces, so piece pb lives in tap pb / (BN / 16)
  auto issue_b = [&](int c, int xi, int stage, int only = -1) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      if ((only < 0 || only == j) && wave + NW * j < BPIECE) {      // wave-uniform
        const float* src = u2 + xi * xi_stride + c * W2_BK + boff[j];
        const int pb = wave + NW * j, tap = pb / (BN / 16), pin = pb - tap * (BN / 16);
        float* slot = stage == 0 ? (tap == 0 ? wsa0 : (tap == 1 ? wsa1 : wsa2)) : (tap == 0 ? wsb0 : (tap == 1 ? wsb1 : wsb2));
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(slot + pin * 256), 16, 0, 0);
      }
    }
  };

  f32x16 yacc[2][2][NJ];
#pragma unroll
  for (int oy = 0; oy < 2; ++oy)
#pragma unroll
    for (int ox = 0; ox < 2; ++ox)
#pragma unroll
      for (int nj = 0; nj < NJ; ++nj)
#pragma unroll
        for (int e = 0; e < 16; ++e) yacc[oy][ox][nj][e] = 0.f;

  // ---- per-lane read addresses (floats) of the 4x4 raw positions of this lane's 2x2 tile -------
  const int li = lane & 31, lh = lane >> 5;
  const int zl = li >> 4, ty = (li >> 2) & 3, tx = li & 3;
  int abase[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int yh = 2 * ty + p, xh = 2 * tx + q;
      const int rl = ((2 * wave + zl) * 10 + yh) * 10 + xh;
      abase[p][q] = (rl ^ ((yh >> 1) & 1)) * 16 + ((lh ^ (((xh >> 2) & 1) | (((yh >> 2) & 1) << 1))) * 4);
    }
  const int bbase = li * 16 + ((lh ^ bswz(li)) * 4);
  const int nchunk = g.Ci / W2_BK;

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                    // every wave is done with the previous chunk's halo and weights
    issue_halo(c);
    issue_b(c, 0, 0);
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      if (WS != 3 || xi == 0) __syncthreads();   // weights of xi (and, at xi = 0, the halo) have landed
      const int stage = WS == 2 ? (xi & 1) : 0;
      if (WS == 2 && xi + 1 < 16) issue_b(c, xi + 1, stage ^ 1);
      const int xy = xi >> 2, xx = xi & 3;
      const int p0 = kBP[xy][0], p1 = kBP[xy][1], q0 = kBP[xx][0], q1 = kBP[xx][1];
      const float s00 = kBS[xy][0] * kBS[xx][0], s01 = kBS[xy][0] * kBS[xx][1];
      const float s10 = kBS[xy][1] * kBS[xx][0], s11 = kBS[xy][1] * kBS[xx][1];
      f32x16 t[NJ];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const float* wtap = stage == 0 ? (a == 0 ? wsa0 : (a == 1 ? wsa1 : wsa2)) : (a == 0 ? wsb0 : (a == 1 ? wsb1 : wsb2));
#pragma unroll
        for (int gk = 0; gk < 2; ++gk) {
          const f32x4 r00 = *reinterpret_cast<const f32x4*>(halo + kx(abase[p0][q0], gk) + a * 1600);
          const f32x4 r01 = *reinterpret_cast<const f32x4*>(halo + kx(abase[p0][q1], gk) + a * 1600);
          const f32x4 r10 = *reinterpret_cast<const f32x4*>(halo + kx(abase[p1][q0], gk) + a * 1600);
          const f32x4 r11 = *reinterpret_cast<const f32x4*>(halo + kx(abase[p1][q1], gk) + a * 1600);
          const f32x4 av = (s00 * r00 + s01 * r01) + (s10 * r10 + s11 * r11);
          f32x4 bf[NJ];
#pragma unroll
          for (int nj = 0; nj < NJ; ++nj)
            bf[nj] = *reinterpret_cast<const f32x4*>(wtap + nj * 32 * 16 + kx(bbase, gk));
#pragma unroll
          for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) {
              if (a == 0 && gk == 0 && e == 0) {
                f32x16 zero;
#pragma unroll
                for (int i = 0; i < 16; ++i) zero[i] = 0.f;
                t[nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bf[nj][e], zero, 0, 0, 0);
              } else {
                t[nj] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[e], bf[nj][e], t[nj], 0, 0, 0);
              }
            }
          }
          __builtin_amdgcn_sched_barrier(0);   // keep one (tap, k-group)'s operands live at a time
        }
        if (WS == 3 && !(xi == 15 && a == 2)) {
          // tap ring (one 4-KB slot per z-tap, NW == 4 and BN == 64: piece j of a wave IS tap j): every wave is
          // done with slot a -> refill it with the next point's tap a, two taps ahead of its use.  A wave's
          // DMAs land in order: all but the newest must be there (the next tap's); none is newer at the end.
          if (xi == 15 && a >= 1) asm volatile("s_waitcnt vmcnt(0)");
          else asm volatile("s_waitcnt vmcnt(1)");
          __builtin_amdgcn_s_barrier();
          if (xi + 1 < 16) issue_b(c, xi + 1, 0, a);
        }
      }
      // output transform on the fly: y[oy][ox] += A^T[oy][xi_y] * A^T[ox][xi_x] * t
      mfma_result_wait<NJ>(t);
#pragma unroll
      for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) {
          const float cf = kAT[oy][xy] * kAT[ox][xx];
          if (cf != 0.f) {
#pragma unroll
            for (int nj = 0; nj < NJ; ++nj) {
              if (cf > 0.f) pk_acc<false>(yacc[oy][ox][nj], t[nj]);   // volatile asm: stays here, in this xi
              else pk_acc<true>(yacc[oy][ox][nj], t[nj]);
            }
          }
        }
      __builtin_amdgcn_sched_barrier(0);   // the adds stay here, in the shadow of this xi's last MFMAs
      if (WS == 1 && xi + 1 < 16) {
        __syncthreads();                  // single weight stage: every wave has read xi's weights
        issue_b(c, xi + 1, 0);
      }
    }
  }

  // ---- epilogue, through LDS ------------------------------------------------------------------
  // The 32x32 accumulator layout gives a lane ONE channel: direct stores are dword stores in 128-B pieces, 128 per
  // lane, and the shortcut-gradient epilogue adds two dword loads per element (the 32 x 64 x 64 data gradients ran
  // 332 us against 271 us for the forward of the same layer).  Each wave turns one (oy, ox) output position of its
  // 32 tiles x 32 NJ channels at a time through a private region of the (now idle) halo buffer -- row pitch + 8 floats --
  // and moves 4 channels = 16 B per lane: a quarter of the memory instructions, whole 128 NJ-byte voxel rows.
  int lhe = lh, lie = li;
  asm volatile("" : "+v"(lhe), "+v"(lie));   // opaque: the epilogue's address math is not hoisted above the main loop
  constexpr int P = 32 * NJ + 8;              // floats per LDS row
  constexpr int Q = 8 * NJ;                   // 4-channel groups per voxel
  constexpr int VPP = 64 / Q;                 // voxels per pass
  static_assert(NW * 32 * P <= W2_HALO, "turn regions fit the halo buffer");
  __syncthreads();                            // every wave is done with the halo
  float* reg = lds + wave * (32 * P);
  const int cq = lane % Q, rs = lane / Q;
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f}, bv = {0.f, 0.f, 0.f, 0.f};
  if (bias) bv = *reinterpret_cast<const f32x4*>(bias + n0 + 4 * cq);
#pragma unroll
  for (int oy = 0; oy < 2; ++oy)
#pragma unroll
    for (int ox = 0; ox < 2; ++ox) {
#pragma unroll
      for (int e = 0; e < 16; ++e)
#pragma unroll
        for (int nj = 0; nj < NJ; ++nj)
          reg[((e & 3) + 8 * (e >> 2) + 4 * lhe) * P + nj * 32 + lie] = yacc[oy][ox][nj][e];
#pragma unroll
      for (int ps = 0; ps < 32 / VPP; ++ps) {
        const int row = ps * VPP + rs;          // 0..31: (zl, ty, tx)
        const int zo = z0 + 2 * wave + (row >> 4);
        const int yo = y0 + 2 * ((row >> 2) & 3) + oy, xo = x0 + 2 * (row & 3) + ox;
        f32x4 v = *reinterpret_cast<const f32x4*>(reg + row * P + 4 * cq);
        if ((zo < g.D) & (yo < g.H) & (xo < g.W)) {
          const long o = ((((long)b * g.D + zo) * g.H + yo) * g.W + xo) * g.No + n0 + 4 * cq;
          v += bv;
          if (add) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(add + o);
            if (gate) {
              const f32x4 gv = *reinterpret_cast<const f32x4*>(gate + o);
#pragma unroll
              for (int j = 0; j < 4; ++j) v[j] += gv[j] > 0.f ? av[j] : 0.f;
            } else {
              v += av;
            }
          }
          *reinterpret_cast<f32x4*>(out + o) = v;
          s1 += v;
          s2 += v * v;
        }
      }
    }
  if (stats) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int o = Q; o < 64; o <<= 1) {
        s1[j] += __shfl_xor(s1[j], o, 64);
        s2[j] += __shfl_xor(s2[j], o, 64);
      }
    __syncthreads();                          // every wave is done with its turn region
    float* red = lds;  // [NW waves][2][BN]
    if (rs == 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        red[(wave * 2 + 0) * BN + 4 * cq + j] = s1[j];
        red[(wave * 2 + 1) * BN + 4 * cq + j] = s2[j];
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, cc = tid - which * BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) v += red[(w * 2 + which) * BN + cc];
      stats[((long)mt * 2 + which) * g.No + n0 + cc] = v;
    }
  }
}

// ==========================================================================================
// v2 of the fused kernel on v_mfma_f32_16x16x4_f32.  Same algorithm and LDS conventions, different
// shape: output tile 8(z) x 8 x 8, a wave owns ONE z-plane = 16 in-plane tiles (16 MFMA rows) x BN
// columns as NB 16x16 blocks.  Per 16x16 block an accumulator is 4 registers, so a wave keeps
// 4 x NB output + NB scratch accumulators = 80 registers for 64 columns (160 in the 32x32 variant):
// the compiler has room to overlap the next operand reads with the MFMAs without spilling, and
//   * one ds_read_b128 per raw position covers all 16 channels of the chunk (lane = (row, k-quad)),
//   * the 10x10x10 halo (64 KB) is double-buffered: the next chunk's halo arrives one DMA piece per
//     wave every other Winograd point, spread over the whole chunk -- no drain at chunk boundaries.
constexpr int V2_TZ = 8;
constexpr int V2_HROWS = 10 * 10 * 10;
constexpr int V2_NPIECE = (V2_HROWS + 15) / 16;          // 63 DMA pieces of 16 rows x 64 B
constexpr int V2_HALO = V2_NPIECE * 256;                 // floats per halo buffer

template <bool NEG>
__device__ __forceinline__ void pk_acc4(f32x4& y, const f32x4& t) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    f32x2 yy = {y[2 * i], y[2 * i + 1]};
    const f32x2 tt = {t[2 * i], t[2 * i + 1]};
    if (NEG) asm volatile("v_pk_add_f32 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]" : "+v"(yy) : "v"(tt));
    else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(yy) : "v"(tt));
    y[2 * i] = yy[0];
    y[2 * i + 1] = yy[1];
  }
}
// v_mfma_f32_16x16x4_f32 (8 passes) result -> VALU read: 11 wait states (operands: see mfma_result_wait)
template <int NB>
__device__ __forceinline__ void mfma16_result_wait(f32x4 (&t)[NB]) {
  if (NB == 4) asm volatile("s_nop 12" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[NB - 1]));
  else asm volatile("s_nop 12" : "+v"(t[0]), "+v"(t[NB - 1]));
}

// weight-row slot swizzle for the (row, k-quad) lane mapping: the four 4-row runs of a 16-lane read
// group carry k-quads {0,0,1,1}, so f(row >> 2) = {0,2,3,1} keeps their 16-B columns distinct
__device__ __forceinline__ int bswz16(int n) { return (0x78 >> (2 * ((n >> 2) & 3))) & 3; }

template <int NB>
__global__ __launch_bounds__(512) void conv_wino2d16_kernel(const float* __restrict__ in, const float* __restrict__ u2,
                                                            const float* __restrict__ bias, float* __restrict__ out,
                                                            float* __restrict__ stats, const float* __restrict__ add,
                                                            const float* __restrict__ gate, const W2dGeom g) {
  constexpr int BN = 16 * NB;
  constexpr int BROWS = 3 * BN;
  constexpr int BST = BROWS * 16;
  constexpr int BPIECE = BROWS / 16;
  constexpr int PB = (BPIECE + 7) / 8;
  // separate LDS objects per buffer (see conv_wino2d_kernel): only then the wait-count pass lets the reads of one
  // buffer run under the outstanding DMAs into the other -- carved from one array the halo prefetch was waited
  // for at every point, which is why this variant first measured no better than the single-buffered one
  __shared__ __attribute__((aligned(1024))) float lds[V2_HALO];     // halo buffer 0 (and the stats scratch)
  __shared__ __attribute__((aligned(1024))) float halo1[V2_HALO];   // halo buffer 1
  __shared__ __attribute__((aligned(1024))) float wst0[BST], wst1[BST];
  __shared__ int htab[V2_NPIECE * 16];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  int L = xcd_remap(blockIdx.x, g.nblk);
  const int n_tile = L % g.n_tiles;
  const int mt = L / g.n_tiles;
  const int b = mt / g.tiles_per_b;
  int r = mt - b * g.tiles_per_b;
  const int txi = r % g.nx; r /= g.nx;
  const int tyi = r % g.ny;
  const int tzi = r / g.ny;
  const int n0 = n_tile * BN;
  const int z0 = tzi * V2_TZ, y0 = tyi * 8, x0 = txi * 8;

  const int pslot = lane & 3;
  for (int rp = tid; rp < V2_NPIECE * 16; rp += 512) {
    const int yh = (rp / 10) % 10;
    const int rl = rp ^ ((yh >> 1) & 1);
    const int zh = rl / 100, xh = rl % 10;
    const int zi = z0 + zh - 1, yi = y0 + yh - 1, xi = x0 + xh - 1;
    const bool v = (rp < V2_HROWS) & (zi >= 0) & (zi < g.D) & (yi >= 0) & (yi < g.H) & (xi >= 0) & (xi < g.W);
    htab[rp] = v ? ((((b * g.D + zi) * g.H + yi) * g.W + xi) * g.Ci) | ((xh >> 2) & 1) | (((yh >> 2) & 1) << 1) : -1;
  }
  int boff[PB];
#pragma unroll
  for (int j = 0; j < PB; ++j) {
    int row = 16 * (wave + 8 * j) + (lane >> 2);
    if (row >= BROWS) row = 0;
    const int a = row / BN, n = row - a * BN;
    boff[j] = (a * g.No + n0 + n) * g.Ci + (pslot ^ bswz16(n)) * 4;
  }
  const long xi_stride = 3L * g.No * g.Ci;

  auto issue_halo_piece = [&](int c, int p, float* buf) __attribute__((always_inline)) {
    const int ho = htab[16 * p + (lane >> 2)];
    const float* src = ho >= 0 ? in + (long)((ho & ~3) + ((pslot ^ (ho & 3)) << 2) + c * W2_BK) : g_w2d_zero;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(buf + p * 256), 16, 0, 0);
  };
  auto issue_b = [&](int c, int xi, int stage) __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < PB; ++j) {
      if (wave + 8 * j < BPIECE) {      // wave-uniform
        const float* src = u2 + xi * xi_stride + c * W2_BK + boff[j];
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)((stage ? wst1 : wst0) + (wave + 8 * j) * 256),
                                         16, 0, 0);
      }
    }
  };

  f32x4 yacc[2][2][NB];
#pragma unroll
  for (int oy = 0; oy < 2; ++oy)
#pragma unroll
    for (int ox = 0; ox < 2; ++ox)
#pragma unroll
      for (int cb = 0; cb < NB; ++cb)
#pragma unroll
        for (int e = 0; e < 4; ++e) yacc[oy][ox][cb][e] = 0.f;

  // lane = (row r = lane & 15 -> in-plane tile (ty, tx), k-quad kq = lane >> 4)
  const int lr = lane & 15, kq = lane >> 4;
  const int ty = lr >> 2, tx = lr & 3;
  int abase[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int yh = 2 * ty + p, xh = 2 * tx + q;
      const int rl = (wave * 10 + yh) * 10 + xh;
      abase[p][q] = (rl ^ ((yh >> 1) & 1)) * 16 + ((kq ^ (((xh >> 2) & 1) | (((yh >> 2) & 1) << 1))) * 4);
    }
  const int bbase = lr * 16 + ((kq ^ bswz16(lr)) * 4);
  const int nchunk = g.Ci / W2_BK;

  __syncthreads();                      // htab complete
#pragma unroll 1
  for (int q = 0; q < 8; ++q)
    if (wave + 8 * q < V2_NPIECE) issue_halo_piece(0, wave + 8 * q, lds);
  issue_b(0, 0, 0);

  // one 16-channel chunk: reads halo buffer hb, prefetches the next chunk's halo into hn (inlined twice with the
  // two buffers in either role, so that each copy names its LDS objects)
  auto chunk = [&](const int c, const float* hb, float* hn) __attribute__((always_inline)) {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) {
      __syncthreads();                  // weights of (c, xi) have landed (at xi = 0 also the rest of halo c)
      if (xi + 1 < 16) issue_b(c, xi + 1, (xi + 1) & 1);
      else if (c + 1 < nchunk) issue_b(c + 1, 0, 0);
      if (c + 1 < nchunk) {             // next chunk's halo: 4 pieces per point, waves 0-3 / 4-7 alternate
        const int w4 = wave - 4 * (xi & 1);
        const int p = xi * 4 + w4;
        if (w4 >= 0 && w4 < 4 && p < V2_NPIECE) issue_halo_piece(c + 1, p, hn);
      }
      const int xy = xi >> 2, xx = xi & 3;
      const int p0 = kBP[xy][0], p1 = kBP[xy][1], q0 = kBP[xx][0], q1 = kBP[xx][1];
      const float s00 = kBS[xy][0] * kBS[xx][0], s01 = kBS[xy][0] * kBS[xx][1];
      const float s10 = kBS[xy][1] * kBS[xx][0], s11 = kBS[xy][1] * kBS[xx][1];
      const float* bs = (xi & 1) ? wst1 : wst0;
      f32x4 t[NB];
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        const f32x4 r00 = *reinterpret_cast<const f32x4*>(hb + abase[p0][q0] + a * 1600);
        const f32x4 r01 = *reinterpret_cast<const f32x4*>(hb + abase[p0][q1] + a * 1600);
        const f32x4 r10 = *reinterpret_cast<const f32x4*>(hb + abase[p1][q0] + a * 1600);
        const f32x4 r11 = *reinterpret_cast<const f32x4*>(hb + abase[p1][q1] + a * 1600);
        f32x4 bf[NB];
#pragma unroll
        for (int cb = 0; cb < NB; ++cb)
          bf[cb] = *reinterpret_cast<const f32x4*>(bs + (a * BN + cb * 16) * 16 + bbase);
        const f32x4 av = (s00 * r00 + s01 * r01) + (s10 * r10 + s11 * r11);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
          for (int cb = 0; cb < NB; ++cb) {
            if (a == 0 && e == 0) {
              const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
              t[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bf[cb][e], zero, 0, 0, 0);
            } else {
              t[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bf[cb][e], t[cb], 0, 0, 0);
            }
          }
        }
      }
      mfma16_result_wait<NB>(t);
#pragma unroll
      for (int oy = 0; oy < 2; ++oy)
#pragma unroll
        for (int ox = 0; ox < 2; ++ox) {
          const float cf = kAT[oy][xy] * kAT[ox][xx];
          if (cf != 0.f) {
#pragma unroll
            for (int cb = 0; cb < NB; ++cb) {
              if (cf > 0.f) pk_acc4<false>(yacc[oy][ox][cb], t[cb]);
              else pk_acc4<true>(yacc[oy][ox][cb], t[cb]);
            }
          }
        }
    }
  };
  for (int c = 0; c < nchunk; c += 2) {
    chunk(c, lds, halo1);
    if (c + 1 < nchunk) chunk(c + 1, halo1, lds);
  }

  // ---- epilogue: D register i of a 16x16 block <-> row 4*kq + i = tile (ty = kq, tx = i), column lr ---
  float s1[NB], s2[NB], bv[NB];
#pragma unroll
  for (int cb = 0; cb < NB; ++cb) {
    s1[cb] = 0.f;
    s2[cb] = 0.f;
    bv[cb] = bias ? bias[n0 + cb * 16 + lr] : 0.f;
  }
  const int zo = z0 + wave;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int oy = 0; oy < 2; ++oy)
#pragma unroll
      for (int ox = 0; ox < 2; ++ox) {
        const int yo = y0 + 2 * kq + oy, xo = x0 + 2 * i + ox;
        if ((zo < g.D) & (yo < g.H) & (xo < g.W)) {
          const long o = ((((long)b * g.D + zo) * g.H + yo) * g.W + xo) * g.No + n0 + lr;
#pragma unroll
          for (int cb = 0; cb < NB; ++cb) {
            float v = yacc[oy][ox][cb][i] + bv[cb];
            if (add) {
              const float av = add[o + cb * 16];
              v += gate ? (gate[o + cb * 16] > 0.f ? av : 0.f) : av;
            }
            out[o + cb * 16] = v;
            s1[cb] += v;
            s2[cb] += v * v;
          }
        }
      }
  }
  if (stats) {
    __syncthreads();
    float* red = lds;  // [8 waves][2][BN]
#pragma unroll
    for (int cb = 0; cb < NB; ++cb) {
      float t1 = s1[cb] + __shfl_xor(s1[cb], 16, 64);
      float t2 = s2[cb] + __shfl_xor(s2[cb], 16, 64);
      t1 += __shfl_xor(t1, 32, 64);
      t2 += __shfl_xor(t2, 32, 64);
      if (kq == 0) {
        red[(wave * 2 + 0) * BN + cb * 16 + lr] = t1;
        red[(wave * 2 + 1) * BN + cb * 16 + lr] = t2;
      }
    }
    __syncthreads();
    if (tid < 2 * BN) {
      const int which = tid / BN, cc = tid - which * BN;
      float v = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) v += red[(w * 2 + which) * BN + cc];
      stats[((long)mt * 2 + which) * g.No + n0 + cc] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// U2[xi][a][n][k] = (G w_a G^T)[xi], G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]
//   blockIdx.y == 0: forward operand   n = co, k = ci
//   blockIdx.y == 1: data-gradient operand n = ci, k = co, all three tap axes flipped
__device__ __forceinline__ void g4_2d(const float a0, const float a1, const float a2, float* r) {
  const float h = 0.5f * (a0 + a2);
  r[0] = a0;
  r[1] = h + 0.5f * a1;
  r[2] = h - 0.5f * a1;
  r[3] = a2;
}

__global__ __launch_bounds__(256) void wino2d_weight_kernel(const float* __restrict__ w, float* __restrict__ uf,
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
  for (int t = 0; t < 27; ++t) (&gw[0][0][0])[t] = src[bwd ? 26 - t : t];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    float p[3][4], u[4][4];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) g4_2d(gw[a][ky][0], gw[a][ky][1], gw[a][ky][2], p[ky]);
#pragma unroll
    for (int x = 0; x < 4; ++x) {
      float col[4];
      g4_2d(p[0][x], p[1][x], p[2][x], col);
#pragma unroll
      for (int y = 0; y < 4; ++y) u[y][x] = col[y];
    }
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) dst[((long)xi * 3 + a) * n + i] = u[xi >> 2][xi & 3];
  }
}

bool w2d_ok(const DramConvDesc* d, int K, int N) {
  if (!d) return false;
  if (d->B < 1 || d->D < 1 || d->H < 1 || d->W < 1) return false;
  if (d->k != 3 || d->stride != 1 || d->dil != 1 || d->pad != 1) return false;
  if (d->Do != d->D || d->Ho != d->H || d->Wo != d->W) return false;
  if (K < 16 || K % 16 != 0 || N < 32 || N % 32 != 0) return false;
  const long long cmax = K > N ? K : N;
  return (long long)d->B * d->D * d->H * d->W * cmax < (1LL << 31);
}

// Kernel variant: 1 = 32x32x2 kernel, 8 waves, 16-deep tiles; 2 = 16x16x4 kernel, 8-deep tiles, double-buffered
// halo; 3 = 32x32x2 kernel as two 4-wave workgroups per CU, 8-deep tiles, weights refilled tap by tap.
// Measured (fwd / dgrad ms, 64->64 @ 2x64x128x128): 2.15 / 2.01, 2.28 / 2.14, 2.13 / 1.94; 64->32 forward
// (32 columns): 1.36, 1.35, 1.39.  So 64-column launches on >= 1024 tiles take variant 3; the others variant 1, or 2 when
// 16-deep tiles would pad the depth more than 8-deep ones or leave the chip under-filled (one workgroup per
// CU: 128 tiles on 256 CUs run at half rate, 256 half-size tiles do not).  DRAM_W2D_V forces one (tests).
int w2d_variant(const DramConvDesc* d, int n_tiles, int BN) {
  if (const char* e = tune_env("DRAM_W2D_V")) {
    const int v = atoi(e);
    if (v >= 1 && v <= 3) return v;
  }
  const int z16 = (d->D + 15) / 16, z8 = (d->D + 7) / 8;
  const long cols = (long)d->B * ((d->H + 7) / 8) * ((d->W + 7) / 8) * n_tiles;
  if (BN == 64 && cols * z8 >= 1024) return 3;       // two 4-wave workgroups per CU: at least two rounds of 512 slots
  if (z16 * 16 != z8 * 8) return 2;
  auto fill = [](long wgs) { return (double)wgs / (double)(((wgs + 255) / 256) * 256); };
  return fill(cols * z16) + 0.05 < fill(cols * z8) ? 2 : 1;
}

W2dGeom make_w2d(const DramConvDesc* d, int K, int N, int BN, int* variant = nullptr) {
  W2dGeom g{};
  g.B = d->B; g.D = d->D; g.H = d->H; g.W = d->W; g.Ci = K; g.No = N;
  g.n_tiles = N / BN;
  const int v = w2d_variant(d, g.n_tiles, BN);
  if (variant) *variant = v;
  const int tz = v == 1 ? W2_TZ16 : V2_TZ;
  g.nz = (g.D + tz - 1) / tz;
  g.ny = (g.H + 7) / 8;
  g.nx = (g.W + 7) / 8;
  g.tiles_per_b = g.nz * g.ny * g.nx;
  g.nblk = g.B * g.tiles_per_b * g.n_tiles;
  return g;
}

int run_w2d(const float* in, const float* u2, const float* bias, const float* add, const float* gate, float* out,
            float* stats, const DramConvDesc* d, int K, int N, hipStream_t s) {
  const int BN = N % 64 == 0 ? 64 : 32;
  int variant = 1;
  const W2dGeom g = make_w2d(d, K, N, BN, &variant);
  // executed: 16 in-plane Winograd points x 3 z-taps = 48 MFMA products per 2x2 output tile and channel pair
  // (a direct conv issues 108) on the padded tile grid; bytes: input, output (+ add, gate) once, weights once
  const double vox_pad = (double)g.B * g.tiles_per_b * (variant == 1 ? W2_TZ16 : V2_TZ) * 64.0;
  const double vox = (double)g.B * g.D * g.H * g.W;
  DramProf prof(DRAM_FAM_CONV_WINO2D, variant * 100 + BN, 2.0 * vox_pad * K * N * 12.0,
                4.0 * (vox * (K + N * (1.0 + (add ? 1 : 0) + (gate ? 1 : 0))) + 48.0 * K * N), s,
                2.0 * vox * K * N * 27.0);
  if (variant == 1) {
    if (BN == 64)
      hipLaunchKernelGGL((conv_wino2d_kernel<2>), dim3(g.nblk), dim3(512), 0, s, in, u2, bias, out, stats, add, gate, g);
    else
      hipLaunchKernelGGL((conv_wino2d_kernel<1>), dim3(g.nblk), dim3(512), 0, s, in, u2, bias, out, stats, add, gate, g);
  } else if (variant == 3) {
    if (BN == 64)
      hipLaunchKernelGGL((conv_wino2d_kernel<2, 4, 3>), dim3(g.nblk), dim3(256), 0, s, in, u2, bias, out, stats, add, gate, g);
    else
      hipLaunchKernelGGL((conv_wino2d_kernel<1, 4, 1>), dim3(g.nblk), dim3(256), 0, s, in, u2, bias, out, stats, add, gate, g);
  } else {
    if (BN == 64)
      hipLaunchKernelGGL((conv_wino2d16_kernel<4>), dim3(g.nblk), dim3(512), 0, s, in, u2, bias, out, stats, add, gate, g);
    else
      hipLaunchKernelGGL((conv_wino2d16_kernel<2>), dim3(g.nblk), dim3(512), 0, s, in, u2, bias, out, stats, add, gate, g);
  }
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

}  // namespace

extern "C" int dram_wino2d_applicable(const DramConvDesc* d) {
  return (d && w2d_ok(d, d->Cin, d->Cout) && w2d_ok(d, d->Cout, d->Cin)) ? 1 : 0;
}

extern "C" int dram_wino2d_num_stat_rows(const DramConvDesc* d) {
  if (!dram_wino2d_applicable(d)) return DRAM_ERR_UNSUPPORTED;
  const W2dGeom g = make_w2d(d, d->Cin, d->Cout, d->Cout % 64 == 0 ? 64 : 32);   // same tiling as the forward launch
  return g.B * g.tiles_per_b;
}

extern "C" int dram_wino2d_pack_weight(const float* w, float* uf, float* ub, int Cout, int Cin, dram_stream_t stream) {
  if (!w || (!uf && !ub) || Cout < 1 || Cin < 1) return DRAM_ERR_BAD_ARG;
  const long n = (long)Cout * Cin;
  DramProf prof(DRAM_FAM_WEIGHT_PACK, 2, 0.0, 4.0 * (double)n * (27.0 + (uf ? 48 : 0) + (ub ? 48 : 0)),
                (hipStream_t)stream);
  hipLaunchKernelGGL(wino2d_weight_kernel, dim3((unsigned)((n + 255) / 256), 2), dim3(256), 0, (hipStream_t)stream, w,
                     uf, ub, Cout, Cin);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_wino2d_conv3d_fwd(const float* x, const float* uf, const float* bias, float* y, float* stats_partial,
                                      const DramConvDesc* d, dram_stream_t stream) {
  if (!x || !uf || !y) return DRAM_ERR_BAD_ARG;
  if (!dram_wino2d_applicable(d)) return DRAM_ERR_UNSUPPORTED;
  return run_w2d(x, uf, bias, nullptr, nullptr, y, stats_partial, d, d->Cin, d->Cout, (hipStream_t)stream);
}

extern "C" int dram_wino2d_conv3d_bwd_data(const float* dy, const float* ub, float* dx, const float* add,
                                           const float* gate, const DramConvDesc* d, dram_stream_t stream) {
  if (!dy || !ub || !dx || (gate && !add)) return DRAM_ERR_BAD_ARG;
  if (!dram_wino2d_applicable(d)) return DRAM_ERR_UNSUPPORTED;
  return run_w2d(dy, ub, nullptr, add, gate, dx, nullptr, d, d->Cout, d->Cin, (hipStream_t)stream);
}
