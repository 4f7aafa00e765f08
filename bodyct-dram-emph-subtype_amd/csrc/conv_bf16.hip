// conv_bf16.hip -- the bf16-storage convolution path (reference: Lightning `--precision bf16`, train.py:46 ->
// torch.autocast(bfloat16) around med3d.py's Conv3d call sites; BASELINE configs[2] and [4]).
//
// Activations and packed weights are bf16 in HBM (NDHWC / [tap][Cout][Cin]), every product is exact in the
// fp32 accumulators of v_mfma_f32_32x32x16_bf16, results are rounded to bf16 once in the epilogue.  At 16x the
// fp32 matrix rate the Winograd forms of the fp32 path are not needed (and their error amplification is avoided):
// these are DIRECT implicit-GEMM kernels.
//
//   conv3_bf16_kernel<NB, EPI>   3x3x3, stride 1, pad = dilation: forward (bias + BatchNorm sums epilogue) and data
//                                gradient (same kernel on the tap-flipped, transposed weights; += add * (gate > 0)
//                                epilogue = the identity-shortcut gradient).  One workgroup (4 waves) = a 4x8x8 block
//                                of the dilation lattice x 32*NB output channels.  Per 32-channel chunk the 6x10x10
//                                input halo (38 KB) is brought into LDS ONCE by LDS-DMA and serves all 27 taps; the
//                                weights stream through two small LDS buffers, three taps (one kx row) at a time.
//                                Both images are XOR-swizzled in 16-B slots so that every ds_read_b128 operand fetch
//                                is conflict-free for the four lane groups of that instruction.
//   wgrad3_bf16_kernel           weight gradient dW[tap][co][ci] = sum_v dy[v][co] x[v + tap][ci]: the contraction runs
//                                over VOXELS, i.e. over the row index of both NDHWC operands, so MFMA operands are
//                                transposed LDS reads (ds_read_b64_tr_b16) of the same halo image + a dy tile.  8 waves;
//                                wave = (32-channel half of a 64-wide co block) x (every 4th tap): 6-7 accumulators.
//                                Partial sums per voxel split -> fp32 slabs -> fixed-order reduce (deterministic).
//   pack / cast kernels          fp32 [Cout][Cin][k^3] -> bf16 wf[tap][Cout][Cin], wb[26 - tap][Cin][Cout]; f32 <-> bf16.
//
// Geometries outside this (the one stride-2 convolution per network, the 1x1x1 convolutions of the Bottleneck
// blocks) are run by the host on the fp32 kernels around cast passes (ops.py) -- a few per cent of the FLOPs.
#include "common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// source of the out-of-range (zero padding) halo granules of the LDS-DMA fills
__device__ __attribute__((aligned(256))) unsigned int g_zero_page[64];

#define GLDS16(src_, dst_)                                                                        \
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src_),         \
                                   (__attribute__((address_space(3))) void*)(dst_), 16, 0, 0)

struct CGeom {
  int B, D, H, W, Cin, Cout, d;
  int Tz, Ty, Tx;      // lattice tiles per (batch, residue class)
  int n_tiles;         // Cout / (32 * NB)
  int nblk;
};

__device__ __forceinline__ void decode_tile(int t, const CGeom& g, int& b, int& rz, int& ry, int& rx, int& lz0,
                                            int& ly0, int& lx0) {
  const int txi = t % g.Tx; t /= g.Tx;
  const int tyi = t % g.Ty; t /= g.Ty;
  const int tzi = t % g.Tz; t /= g.Tz;
  rx = t % g.d; t /= g.d;
  ry = t % g.d; t /= g.d;
  rz = t % g.d;
  b = t / g.d;
  lz0 = tzi * 4; ly0 = tyi * 8; lx0 = txi * 8;
}

constexpr int HALO_GRAN = 2560;   // 6 x 10 x 10 voxels x 4 slots = 2400 16-B granules, padded to 10 x 256

// ---------------------------------------------------------------------------------------------------------
// forward / data gradient
template <int NB, int EPI>
__global__ __launch_bounds__(256, 2) void conv3_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const float* __restrict__ bias,
                                                            const bf16_t* __restrict__ add,
                                                            const bf16_t* __restrict__ gate, bf16_t* __restrict__ y,
                                                            float* __restrict__ stats, const CGeom g) {
  constexpr int WROWS = 3 * 32 * NB;                       // weight rows (kx, co) per group
  constexpr int WROUNDS = (WROWS * 4 + 255) / 256;         // DMA rounds per group
  // one LDS object per buffer: the wait-count pass only keeps a pending DMA out of the way of reads that provably
  // touch another object
  __shared__ __attribute__((aligned(1024))) unsigned char halo[HALO_GRAN * 16];
  __shared__ __attribute__((aligned(1024))) unsigned char wb0[WROUNDS * 4096];
  __shared__ __attribute__((aligned(1024))) unsigned char wb1[WROUNDS * 4096];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int L = xcd_remap(blockIdx.x, g.nblk);
  const int nt = L % g.n_tiles;
  int b, rz, ry, rx, lz0, ly0, lx0;
  decode_tile(L / g.n_tiles, g, b, rz, ry, rx, lz0, ly0, lx0);
  const int co0 = nt * 32 * NB;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page);

  // halo DMA sources: granule p = round * 256 + tid -> voxel p / 4, physical slot p % 4 holding the logical slot
  // (p % 4) ^ ((hx >> 1) & 3)
  int hoff[10];
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const int p = i * 256 + tid;
    const int hv = p >> 2, ps = p & 3;
    const int hz = hv / 100, hy = (hv / 10) % 10, hx = hv % 10;
    const int gz = rz + g.d * (lz0 + hz - 1), gy = ry + g.d * (ly0 + hy - 1), gx = rx + g.d * (lx0 + hx - 1);
    const bool ok = (hv < 600) & (gz >= 0) & (gz < g.D) & (gy >= 0) & (gy < g.H) & (gx >= 0) & (gx < g.W);
    hoff[i] = ok ? (int)(((((long)b * g.D + gz) * g.H + gy) * g.W + gx) * g.Cin) + (ps ^ ((hx >> 1) & 3)) * 8 : -1;
  }
  // weight DMA sources: granule q -> row q / 4 = kx * 32 NB + co, physical slot q % 4 = logical ^ ((co >> 2) & 3)
  int woff[WROUNDS];
#pragma unroll
  for (int r = 0; r < WROUNDS; ++r) {
    const int q = r * 256 + tid;
    const int row = q >> 2, ps = q & 3;
    const int kx = row / (32 * NB), co = row % (32 * NB);
    woff[r] = row < WROWS ? (kx * g.Cout + co0 + co) * g.Cin + (ps ^ ((co >> 2) & 3)) * 8 : -1;
  }
  auto issue_halo = [&](int c) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 10; ++i) {
      const unsigned char* src = hoff[i] >= 0 ? reinterpret_cast<const unsigned char*>(x + hoff[i] + c * 32) : zero;
      GLDS16(src, halo + i * 4096 + wave * 1024);
    }
  };
  auto issue_w = [&](int c, int gi, unsigned char* buf) __attribute__((always_inline)) {
    const bf16_t* wg = w + (long)(3 * gi) * g.Cout * g.Cin + c * 32;
#pragma unroll
    for (int r = 0; r < WROUNDS; ++r) {
      const unsigned char* src = woff[r] >= 0 ? reinterpret_cast<const unsigned char*>(wg + woff[r]) : zero;
      GLDS16(src, buf + r * 4096 + wave * 1024);
    }
  };

  f32x16 acc[2][NB];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nb][e] = 0.f;

  // operand read addresses (bytes): A row li of M block mi = halo voxel (wave + kz, 4 mi + li / 8 + ky, li % 8 + kx)
  const int xl = li & 7;
  int abase[2], akey[3];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) abase[mi] = ((wave * 10 + 4 * mi + (li >> 3)) * 10 + xl) * 64;
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) akey[kx] = ((xl + kx) >> 1) & 3;
  const int bkey = (li >> 2) & 3;

  const int nchunk = g.Cin / 32;
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                       // every wave is done with the halo and both weight buffers of the last chunk
    issue_halo(c);
    issue_w(c, 0, wb0);
#pragma unroll
    for (int gi = 0; gi < 9; ++gi) {
      __syncthreads();                     // group gi's weights (gi == 0: and the halo) have landed; buffer (gi+1)&1 is free
      if (gi < 8) issue_w(c, gi + 1, ((gi + 1) & 1) ? wb1 : wb0);
      const unsigned char* wl = (gi & 1) ? wb1 : wb0;
      const int tapo = ((gi / 3) * 100 + (gi % 3) * 10) * 64;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          bf16x8 af[2], bfr[NB];
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
            af[mi] = *reinterpret_cast<const bf16x8*>(halo + abase[mi] + tapo + kx * 64 + (((2 * j + lh) ^ akey[kx]) << 4));
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            bfr[nb] = *reinterpret_cast<const bf16x8*>(wl + ((kx * 32 * NB + nb * 32 + li) << 6) + (((2 * j + lh) ^ bkey) << 4));
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
              acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[mi], bfr[nb], acc[mi][nb], 0, 0, 0);
        }
    }
  }

  // epilogue: D[row][col]: col = li (channel), row = (e & 3) + 8 (e >> 2) + 4 lh within the 32-row block
  float s1[NB], s2[NB], bv[NB];
#pragma unroll
  for (int nb = 0; nb < NB; ++nb) {
    s1[nb] = 0.f;
    s2[nb] = 0.f;
    bv[nb] = (EPI == 0 && bias) ? bias[co0 + nb * 32 + li] : 0.f;
  }
  const int gz = rz + g.d * (lz0 + wave);
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
      const int gy = ry + g.d * (ly0 + 4 * mi + (row >> 3)), gx = rx + g.d * (lx0 + (row & 7));
      if ((gz < g.D) & (gy < g.H) & (gx < g.W)) {
        const long o = ((((long)b * g.D + gz) * g.H + gy) * g.W + gx) * g.Cout + co0 + li;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
          float v = acc[mi][nb][e] + bv[nb];
          if (EPI == 1 && add) {
            const float av = bf16_to_f32(add[o + nb * 32]);
            v += gate ? (bf16_to_f32(gate[o + nb * 32]) > 0.f ? av : 0.f) : av;
          }
          const bf16_t h = f32_to_bf16(v);
          y[o + nb * 32] = h;
          if (EPI == 0) {                  // BatchNorm sums of the values the next pass will read (the rounded ones)
            const float vr = bf16_to_f32(h);
            s1[nb] += vr;
            s2[nb] += vr * vr;
          }
        }
      }
    }
  if (EPI == 0 && stats) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(halo);   // [4 waves][2][32 NB]
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
      const float t1 = s1[nb] + __shfl_xor(s1[nb], 32, 64);
      const float t2 = s2[nb] + __shfl_xor(s2[nb], 32, 64);
      if (lh == 0) {
        red[(wave * 2 + 0) * 32 * NB + nb * 32 + li] = t1;
        red[(wave * 2 + 1) * 32 * NB + nb * 32 + li] = t2;
      }
    }
    __syncthreads();
    if (tid < 2 * 32 * NB) {
      const int which = tid / (32 * NB), cc = tid % (32 * NB);
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) v += red[(wv * 2 + which) * 32 * NB + cc];
      stats[((long)(L / g.n_tiles) * 2 + which) * g.Cout + co0 + cc] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient
struct WGeom {
  int B, D, H, W, Cin, Cout, d;
  int Tz, Ty, Tx;
  int ntile;           // spatial tiles
  int nsplit;          // voxel splits (slabs)
  int ci_blocks, co_blocks, npairs;
  int nblk;
};

__global__ __launch_bounds__(512) void wgrad3_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          float* __restrict__ slab, const WGeom g) {
  __shared__ __attribute__((aligned(1024))) unsigned char halo0[HALO_GRAN * 16];
  __shared__ __attribute__((aligned(1024))) unsigned char halo1[HALO_GRAN * 16];
  __shared__ __attribute__((aligned(1024))) unsigned char dyt0[2048 * 16];      // [256 voxels][64 co] bf16
  __shared__ __attribute__((aligned(1024))) unsigned char dyt1[2048 * 16];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, g.nblk);
  const int pair = L % g.npairs, split = L / g.npairs;
  const int cib = pair % g.ci_blocks, cob = pair / g.ci_blocks;
  const unsigned char* zero = reinterpret_cast<const unsigned char*>(g_zero_page);
  const CGeom cg{g.B, g.D, g.H, g.W, g.Cin, g.Cout, g.d, g.Tz, g.Ty, g.Tx, 1, 0};

  auto issue_tile = [&](int t, unsigned char* hb, unsigned char* db) __attribute__((always_inline)) {
    int b, rz, ry, rx, lz0, ly0, lx0;
    decode_tile(t, cg, b, rz, ry, rx, lz0, ly0, lx0);
#pragma unroll
    for (int i = 0; i < 5; ++i) {              // halo: 2560 granules / 512 threads
      const int p = i * 512 + tid;
      const int hv = p >> 2, ps = p & 3;
      const int hz = hv / 100, hy = (hv / 10) % 10, hx = hv % 10;
      const int gz = rz + g.d * (lz0 + hz - 1), gy = ry + g.d * (ly0 + hy - 1), gx = rx + g.d * (lx0 + hx - 1);
      const bool ok = (hv < 600) & (gz >= 0) & (gz < g.D) & (gy >= 0) & (gy < g.H) & (gx >= 0) & (gx < g.W);
      const long off = ((((long)b * g.D + gz) * g.H + gy) * g.W + gx) * g.Cin + cib * 32 + (ps ^ ((hx >> 1) & 3)) * 8;
      const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(x + off) : zero;
      GLDS16(src, hb + i * 8192 + wave * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {              // dy tile: 256 voxels x 8 slots (64 co); slot ^= 4 * ((v >> 1) & 1)
      const int q = i * 512 + tid;
      const int v = q >> 3, ps = q & 7;
      const int slot = ps ^ (((v >> 1) & 1) << 2);
      const int gz = rz + g.d * (lz0 + (v >> 6)), gy = ry + g.d * (ly0 + ((v >> 3) & 7)), gx = rx + g.d * (lx0 + (v & 7));
      const int ch = cob * 64 + slot * 8;
      const bool ok = (gz < g.D) & (gy < g.H) & (gx < g.W) & (ch < g.Cout);
      const long off = ((((long)b * g.D + gz) * g.H + gy) * g.W + gx) * g.Cout + ch;
      const unsigned char* src = ok ? reinterpret_cast<const unsigned char*>(dy + off) : zero;
      GLDS16(src, db + i * 8192 + wave * 1024);
    }
  };

  // this wave: co half cb of the block's 64, taps t0, t0 + 4, ... (7 of them; 6 for t0 == 3)
  const int cb = wave & 1, t0 = wave >> 1;
  const int ntap = t0 < 3 ? 7 : 6;
  f32x16 acc[7];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

  // transposed-read addresses: 16-lane group g4 = lane >> 4 (bit 0: column half, bit 1: k half), lane 4 q + p of a
  // group addresses row q, the 8-byte piece p of the group's 16 columns
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = g4 >> 1;
  const int cslot = (g4 & 1) * 2 + (p >> 1), cbyte = (p & 1) * 8;
  // A (dy): voxel (zt, 2 yp + h, 4 s + q) of the tile -> byte v * 128 + ((4 cb + cslot) ^ 4 ((v >> 1) & 1)) * 16 + cbyte;
  // with x = 4 s + q: (v >> 1) & 1 = (q >> 1) & 1 for both s
  int aoff[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const int v = h * 8 + 4 * s + q;
    aoff[s] = v * 128 + (((4 * cb + cslot) ^ (((v >> 1) & 1) << 2)) << 4) + cbyte;
  }
  // B (x halo): voxel (zt + kz, 2 yp + h + ky, 4 s + q + kx) -> byte hv * 64 + (cslot ^ ((hx >> 1) & 3)) * 16 + cbyte
  int boff[7][2];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int tap = (t0 + 4 * i) < 27 ? t0 + 4 * i : 0;
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int hx = 4 * s + q + kx;
      boff[i][s] = ((kz * 10 + h + ky) * 10 + hx) * 64 + ((cslot ^ ((hx >> 1) & 3)) << 4) + cbyte;
    }
  }
  auto tr8 = [&](const unsigned char* base, int o0, int o1) __attribute__((always_inline)) {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o0));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7));
  };

  int t = split;
  if (t < g.ntile) issue_tile(t, halo0, dyt0);
  int buf = 0;
  for (; t < g.ntile; t += g.nsplit, buf ^= 1) {
    __syncthreads();                                     // tile t has landed; the other buffer pair is free
    if (t + g.nsplit < g.ntile) {
      if (buf == 0) issue_tile(t + g.nsplit, halo1, dyt1);
      else issue_tile(t + g.nsplit, halo0, dyt0);
    }
    const unsigned char* hb = buf == 0 ? halo0 : halo1;
    const unsigned char* db = buf == 0 ? dyt0 : dyt1;
#pragma unroll 2
    for (int ks = 0; ks < 16; ++ks) {
      const int zt = ks >> 2, yp = ks & 3;
      const unsigned char* da = db + (zt * 64 + yp * 16) * 128;            // voxel (zt, 2 yp, 0)
      const unsigned char* ha = hb + (zt * 100 + yp * 20) * 64;            // halo voxel (zt, 2 yp, 0)
      const bf16x8 af = tr8(da, aoff[0], aoff[1]);
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        if (i < ntap) {
          const bf16x8 bfr = tr8(ha, boff[i][0], boff[i][1]);
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[i], 0, 0, 0);
        }
      }
    }
  }

  // slab[split][pair][tap][co 64][ci 32]
  const int li = lane & 31, lh = lane >> 5;
  float* sb = slab + ((long)split * g.npairs + pair) * 27 * 2048;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    if (i < ntap) {
      const int tap = t0 + 4 * i;
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
        sb[(tap * 64 + cb * 32 + row) * 32 + li] = acc[i][e];
      }
    }
  }
}

// dw[co][ci][tap] = sum_split slab[split][pair(co / 64, ci / 32)][tap][co % 64][ci % 32], fixed order
__global__ void wgrad3_bf16_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin,
                                          int ci_blocks, int npairs, int nsplit) {
  const long n = (long)Cout * Cin * 27;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    long r = i / Cin;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout);
    const int pair = (co >> 6) * ci_blocks + (ci >> 5);
    const float* s = slab + ((long)pair * 27 + tap) * 2048 + (co & 63) * 32 + (ci & 31);
    float a = 0.f;
    for (int k = 0; k < nsplit; ++k) a += s[(long)k * npairs * 27 * 2048];
    dw[((long)co * Cin + ci) * 27 + tap] = a;
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight packing and casts
__global__ void pack_weight_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf, bf16_t* __restrict__ wb,
                                        int Cout, int Cin, int taps) {
  const long n = (long)Cout * Cin * taps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    // i enumerates the OUTPUT wf[tap][co][ci] (coalesced writes)
    const int ci = (int)(i % Cin);
    long r = i / Cin;
    const int co = (int)(r % Cout);
    const int tap = (int)(r / Cout);
    const bf16_t v = f32_to_bf16(w[((long)co * Cin + ci) * taps + tap]);
    if (wf) wf[i] = v;
    if (wb) wb[((long)(taps - 1 - tap) * Cin + ci) * Cout + co] = v;
  }
}

__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n4, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    st4<bf16_t>(dst, 4 * i, ld4<float>(src, 4 * i));
  for (long i = 4 * n4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = f32_to_bf16(src[i]);
}
__global__ void cast_bf16_f32_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n4, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x)
    st4<float>(dst, 4 * i, ld4<bf16_t>(src, 4 * i));
  for (long i = 4 * n4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = bf16_to_f32(src[i]);
}

inline int ew_grid(long total) {
  long b = (total + 255) / 256;
  return (int)(b > 8192 ? 8192 : (b < 1 ? 1 : b));
}

bool geom_ok(const DramConvDesc* d) {
  if (!d) return false;
  if (d->B < 1 || d->D < 1 || d->H < 1 || d->W < 1) return false;
  if (d->k != 3 || d->stride != 1 || d->dil < 1 || d->pad != d->dil) return false;
  if (d->Do != d->D || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cin < 32 || d->Cout < 32 || d->Cin % 32 != 0 || d->Cout % 32 != 0) return false;
  const long long vox = (long long)d->B * d->D * d->H * d->W;
  const long long cmax = d->Cin > d->Cout ? d->Cin : d->Cout;
  if (vox * cmax >= (1LL << 31)) return false;                          // 32-bit element offsets
  if ((long long)d->Cout * d->Cin * 27 >= (1LL << 31)) return false;
  return true;
}

// forward-sense descriptor -> lattice tiling of a tensor with the spatial extent (D, H, W)
void fill_tiles(const DramConvDesc* d, int& Tz, int& Ty, int& Tx, long long& ntile) {
  auto tiles = [&](int n, int per) { return ((n + d->dil - 1) / d->dil + per - 1) / per; };
  Tz = tiles(d->D, 4); Ty = tiles(d->H, 8); Tx = tiles(d->W, 8);
  ntile = (long long)d->B * d->dil * d->dil * d->dil * Tz * Ty * Tx;
}

// N tile: 64 columns (2 blocks); 32 when the channel count is not a multiple of 64
int pick_nb(int cout) { return cout % 64 == 0 ? 2 : 1; }

int launch_conv(const bf16_t* x, const bf16_t* w, const float* bias, const bf16_t* add, const bf16_t* gate, bf16_t* y,
                float* stats, const DramConvDesc* d, int cin, int cout, int epi, hipStream_t s) {
  CGeom g{};
  g.B = d->B; g.D = d->D; g.H = d->H; g.W = d->W; g.Cin = cin; g.Cout = cout; g.d = d->dil;
  long long ntile;
  fill_tiles(d, g.Tz, g.Ty, g.Tx, ntile);
  const int nb = pick_nb(cout);
  g.n_tiles = cout / (32 * nb);
  const long long nblk = ntile * g.n_tiles;
  if (nblk >= (1LL << 31)) return DRAM_ERR_UNSUPPORTED;
  g.nblk = (int)nblk;
  const double vox = (double)d->B * d->D * d->H * d->W;
  const double flops = 2.0 * (double)ntile * 256.0 * 27.0 * cin * cout;      // executed, padded tiles included
  DramProf prof(DRAM_FAM_CONV_BF16, epi * 2 + (nb - 1), flops,
                2.0 * (vox * (cin + cout * (1.0 + (add ? 1 : 0) + (gate ? 1 : 0))) + 27.0 * cin * cout), s,
                2.0 * vox * 27.0 * cin * cout);
#define LAUNCH_(NB_, EPI_)                                                                                         \
  hipLaunchKernelGGL((conv3_bf16_kernel<NB_, EPI_>), dim3(g.nblk), dim3(256), 0, s, x, w, bias, add, gate, y, stats, g)
  if (epi == 0) { if (nb == 2) LAUNCH_(2, 0); else LAUNCH_(1, 0); }
  else          { if (nb == 2) LAUNCH_(2, 1); else LAUNCH_(1, 1); }
#undef LAUNCH_
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

void plan_wgrad(const DramConvDesc* d, WGeom& g) {
  g.B = d->B; g.D = d->D; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Cout = d->Cout; g.d = d->dil;
  long long ntile;
  fill_tiles(d, g.Tz, g.Ty, g.Tx, ntile);
  g.ntile = (int)ntile;
  g.ci_blocks = d->Cin / 32;
  g.co_blocks = (d->Cout + 63) / 64;
  g.npairs = g.ci_blocks * g.co_blocks;
  // ~4 workgroups per CU in total, every split at least 2 tiles deep (double buffering), at most 256 slabs
  int ns = (1024 + g.npairs - 1) / g.npairs;
  if (ns > g.ntile / 2) ns = g.ntile / 2;
  if (ns > 256) ns = 256;
  if (ns < 1) ns = 1;
  g.nsplit = ns;
  g.nblk = g.npairs * g.nsplit;
}

}  // namespace

extern "C" int dram_conv_bf16_supported(const DramConvDesc* d) { return geom_ok(d) ? 1 : 0; }

extern "C" int dram_conv_bf16_num_stat_rows(const DramConvDesc* d) {
  if (!geom_ok(d)) return DRAM_ERR_UNSUPPORTED;
  int Tz, Ty, Tx;
  long long ntile;
  fill_tiles(d, Tz, Ty, Tx, ntile);
  return ntile < (1LL << 31) ? (int)ntile : DRAM_ERR_UNSUPPORTED;
}

extern "C" int dram_pack_conv_weight_bf16(const float* w, void* wf, void* wb, int Cout, int Cin, int taps,
                                          dram_stream_t stream) {
  if (!w || (!wf && !wb) || Cout < 1 || Cin < 1 || taps < 1) return DRAM_ERR_BAD_ARG;
  const long n = (long)Cout * Cin * taps;
  DramProf prof(DRAM_FAM_WEIGHT_PACK, 8, 0.0, (double)n * (4.0 + 2.0 * ((wf ? 1 : 0) + (wb ? 1 : 0))), (hipStream_t)stream);
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wf,
                     (bf16_t*)wb, Cout, Cin, taps);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_cast_f32_to_bf16(const float* src, void* dst, long long n, dram_stream_t stream) {
  if (!src || !dst || n < 1) return DRAM_ERR_BAD_ARG;
  const long n4 = ((((uintptr_t)src & 15) | ((uintptr_t)dst & 7)) == 0) ? n / 4 : 0;
  DramProf prof(DRAM_FAM_BN, 7, 0.0, 6.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(ew_grid(n4 > 0 ? n4 : n)), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, n4, (long)n);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_cast_bf16_to_f32(const void* src, float* dst, long long n, dram_stream_t stream) {
  if (!src || !dst || n < 1) return DRAM_ERR_BAD_ARG;
  const long n4 = ((((uintptr_t)dst & 15) | ((uintptr_t)src & 7)) == 0) ? n / 4 : 0;
  DramProf prof(DRAM_FAM_BN, 8, 0.0, 6.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3(ew_grid(n4 > 0 ? n4 : n)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, dst, n4, (long)n);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_conv3d_fwd_bf16(const void* x, const void* wf, const float* bias, void* y, float* stats_partial,
                                    const DramConvDesc* d, dram_stream_t stream) {
  if (!x || !wf || !y) return DRAM_ERR_BAD_ARG;
  if (!geom_ok(d)) return DRAM_ERR_UNSUPPORTED;
  return launch_conv((const bf16_t*)x, (const bf16_t*)wf, bias, nullptr, nullptr, (bf16_t*)y, stats_partial, d, d->Cin,
                     d->Cout, 0, (hipStream_t)stream);
}

extern "C" int dram_conv3d_bwd_data_bf16(const void* dy, const void* wb, void* dx, const void* add, const void* gate,
                                         const DramConvDesc* d, dram_stream_t stream) {
  if (!dy || !wb || !dx || (gate && !add)) return DRAM_ERR_BAD_ARG;
  if (!geom_ok(d)) return DRAM_ERR_UNSUPPORTED;
  return launch_conv((const bf16_t*)dy, (const bf16_t*)wb, nullptr, (const bf16_t*)add, (const bf16_t*)gate, (bf16_t*)dx,
                     nullptr, d, d->Cout, d->Cin, 1, (hipStream_t)stream);
}

extern "C" size_t dram_conv3d_bwd_weight_bf16_workspace(const DramConvDesc* d) {
  if (!geom_ok(d)) return 0;
  WGeom g{};
  plan_wgrad(d, g);
  return (size_t)g.nsplit * g.npairs * 27 * 2048 * sizeof(float);
}

extern "C" int dram_conv3d_bwd_weight_bf16(const void* x, const void* dy, float* dw, const DramConvDesc* d,
                                           void* workspace, size_t workspace_bytes, dram_stream_t stream) {
  if (!x || !dy || !dw) return DRAM_ERR_BAD_ARG;
  if (!geom_ok(d)) return DRAM_ERR_UNSUPPORTED;
  WGeom g{};
  plan_wgrad(d, g);
  const size_t need = (size_t)g.nsplit * g.npairs * 27 * 2048 * sizeof(float);
  if (!workspace || workspace_bytes < need) return DRAM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const double vox = (double)d->B * d->D * d->H * d->W;
  {
    DramProf prof(DRAM_FAM_WGRAD_BF16, 0, 2.0 * (double)g.ntile * 256.0 * 27.0 * g.ci_blocks * 32.0 * g.co_blocks * 64.0,
                  2.0 * vox * (d->Cin + d->Cout) + 4.0 * 27.0 * d->Cin * d->Cout, s, 2.0 * vox * 27.0 * d->Cin * d->Cout);
    hipLaunchKernelGGL(wgrad3_bf16_kernel, dim3(g.nblk), dim3(512), 0, s, (const bf16_t*)x, (const bf16_t*)dy,
                       (float*)workspace, g);
    DRAM_LAUNCH_CHECK();
  }
  const long n = (long)d->Cout * d->Cin * 27;
  DramProf prof(DRAM_FAM_WGRAD_BF16, 1, 0.0, 4.0 * (double)n * (g.nsplit + 1), s);
  hipLaunchKernelGGL(wgrad3_bf16_reduce_kernel, dim3(ew_grid(n)), dim3(256), 0, s, (const float*)workspace, dw, d->Cout,
                     d->Cin, g.ci_blocks, g.npairs, g.nsplit);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
