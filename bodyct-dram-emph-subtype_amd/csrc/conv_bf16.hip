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
//                                weights stream through a 3-slot LDS ring, three taps (one kx row) per slot, two slots ahead.
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
#include <stdlib.h>
#include <string.h>
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

// Ablation switch of the tuning builds (tools/conv_bf16_ablate.sh; 0 = the product): 1 no weight DMA, 2 no halo DMA,
// 3 no LDS operand reads, 4 no MFMAs, 5 no epilogue stores, 6 no barriers.  Results are garbage in those builds.
// "s_waitcnt vmcnt(0)" as the BUILTIN (gfx9 encoding: lgkmcnt 15, expcnt 7, vmcnt 0), not as inline asm: the wait-count
// pass reads the builtin and knows that no LDS-DMA is outstanding behind it.  Behind an asm wait it still believed the
// previous step's DMA pending and guarded the first operand read of every step with a wait that also covered part of
// the look-ahead DMA just issued (wgrad3z_bf16_kernel: "s_waitcnt vmcnt(2)" with three new DMAs in flight).
#define WAIT_VM0()                         \
  do {                                     \
    __builtin_amdgcn_s_waitcnt(0x0F70);    \
    asm volatile("" ::: "memory");         \
  } while (0)

#ifndef DRAM_BF16_ABL
#define DRAM_BF16_ABL 0
#endif
constexpr int HALO_GRAN = 2560;   // 6 x 10 x 10 voxels x 4 slots = 2400 16-B granules, padded to 10 x 256

// LDS-DMA through a buffer descriptor: the per-lane source is a 32-bit BYTE offset held in a register for the whole
// kernel, an out-of-range offset (0xFFFFFFFF = zero padding of the halo / unused lanes of a piece) reads zeros in
// hardware, and the chunk / tap-group base moves by scalar arithmetic on the descriptor -- one instruction per piece
// instead of a 64-bit select + add per piece.
#define BUFLDS16(rsrc_, voff_, dst_)                                                                               \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc_), (__attribute__((address_space(3))) void*)(dst_), 16, (voff_), 0, 0, 0)
constexpr int BUF_FLAGS = 0x00020000;   // raw buffer, 32-bit data format
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* base, long off_bytes, long total_bytes) {
  long left = total_bytes - off_bytes;
  if (left < 0 || off_bytes < 0) { left = 0; off_bytes = 0; }       // a window wholly outside the tensor: empty descriptor
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(reinterpret_cast<const char*>(base)) + off_bytes, 0,
                                           (int)(left > 0xffffffffL ? 0xffffffffL : left), BUF_FLAGS);
}

// ---------------------------------------------------------------------------------------------------------
// forward / data gradient.  NW waves = NW z-slices of 8 x 8 voxels (NW = 4: 256 voxels, two workgroups per CU;
// NW = 8: 512 voxels, one workgroup per CU -- half the DMA pieces per MFMA, for lattices at least 8 deep).
template <int NB, int EPI, int NW>
__global__ __launch_bounds__(64 * NW, 2) void conv3_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                                const float* __restrict__ bias,
                                                                const bf16_t* __restrict__ add,
                                                                const bf16_t* __restrict__ gate, bf16_t* __restrict__ y,
                                                                float* __restrict__ stats, const CGeom g) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass only needs the launch stub: the buffer-descriptor builtins in
                                      // this TEMPLATE's body made it drop the stub's definition)
  constexpr int NT = 64 * NW;                              // threads
  constexpr int HV = (NW + 2) * 100;                       // halo voxels
  constexpr int HROUNDS = (HV * 4 + NT - 1) / NT;          // DMA rounds of the halo (pieces per wave)
  constexpr int WROWS = 3 * 32 * NB;                       // weight rows (kx, co) per group
  constexpr int WROUNDS = (WROWS * 4 + NT - 1) / NT;       // DMA rounds per group
  // one LDS object per buffer: the wait-count pass only keeps a pending DMA out of the way of reads that provably
  // touch another object
  __shared__ __attribute__((aligned(1024))) unsigned char halo[HROUNDS * NT * 16];
  __shared__ __attribute__((aligned(1024))) unsigned char wb0[WROUNDS * NT * 16];   // weight ring: group g lives in
  __shared__ __attribute__((aligned(1024))) unsigned char wb1[WROUNDS * NT * 16];   // slot g % 3 and is fetched two
  __shared__ __attribute__((aligned(1024))) unsigned char wb2[WROUNDS * NT * 16];   // groups ahead of its use

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int L = xcd_remap(blockIdx.x, g.nblk);
  const int nt = L % g.n_tiles;
  int b, rz, ry, rx, lz0, ly0, lx0;
  decode_tile(L / g.n_tiles, g, b, rz, ry, rx, lz0, ly0, lx0);
  lz0 = (lz0 / 4) * NW;
  const int co0 = nt * 32 * NB;
  const long xbytes = (long)g.B * g.D * g.H * g.W * g.Cin * 2, wbytes = 27L * g.Cout * g.Cin * 2;

  // halo DMA sources: granule p = round * NT + tid -> voxel p / 4, physical slot p % 4 holding the logical slot
  // (p % 4) ^ ((hx >> 1) & 3)
  // (offsets are relative to the first in-volume z plane the halo touches -- the descriptor starts there --, so
  // only the halo's own window has to fit 32 bits, not the tensor: us1.0 of ResNet-50 at 256x512x512 reads 4.8 GB)
  const int gz_first = (rz + g.d * (lz0 - 1)) < 0 ? rz : rz + g.d * (lz0 - 1);
  const long xbase = (((long)b * g.D + (gz_first < g.D ? gz_first : 0)) * g.H * g.W) * g.Cin * 2;
  unsigned hoff[HROUNDS];
#pragma unroll
  for (int i = 0; i < HROUNDS; ++i) {
    const int p = i * NT + tid;
    const int hv = p >> 2, ps = p & 3;
    const int hz = hv / 100, hy = (hv / 10) % 10, hx = hv % 10;
    const int gz = rz + g.d * (lz0 + hz - 1), gy = ry + g.d * (ly0 + hy - 1), gx = rx + g.d * (lx0 + hx - 1);
    const bool ok = (hv < HV) & (gz >= 0) & (gz < g.D) & (gy >= 0) & (gy < g.H) & (gx >= 0) & (gx < g.W);
    const long e = ((((long)(gz - gz_first)) * g.H + gy) * g.W + gx) * g.Cin + (ps ^ ((hx >> 1) & 3)) * 8;
    hoff[i] = ok ? (unsigned)(e * 2) : 0xffffffffu;
  }
  // weight DMA sources: granule q -> row q / 4 = kx * 32 NB + co, physical slot q % 4 = logical ^ ((co >> 2) & 3)
  unsigned woff[WROUNDS];
#pragma unroll
  for (int r = 0; r < WROUNDS; ++r) {
    const int q = r * NT + tid;
    const int row = q >> 2, ps = q & 3;
    const int kx = row / (32 * NB), co = row % (32 * NB);
    woff[r] = row < WROWS ? (unsigned)((((long)kx * g.Cout + co0 + co) * g.Cin + (ps ^ ((co >> 2) & 3)) * 8) * 2) : 0xffffffffu;
  }
  auto issue_halo = [&](int c) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = make_rsrc(x, xbase + (long)c * 64, xbytes);
    if (DRAM_BF16_ABL == 2) return;
#pragma unroll
    for (int i = 0; i < HROUNDS; ++i) BUFLDS16(rs, hoff[i], halo + i * (NT * 16) + wave * 1024);
  };
  auto issue_w = [&](int c, int gi, unsigned char* buf) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t rs = DRAM_BF16_ABL == 8 ? make_rsrc(w, 0, wbytes)
                                                          : make_rsrc(w, ((long)(3 * gi) * g.Cout * g.Cin + c * 32) * 2, wbytes);
    if (DRAM_BF16_ABL == 1) return;
#pragma unroll
    for (int r = 0; r < WROUNDS; ++r) BUFLDS16(rs, woff[r], buf + r * (NT * 16) + wave * 1024);
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

  // Schedule.  The weights of tap group gi (3 taps = one kx row) are issued TWO groups ahead of their use into a
  // 3-slot ring.  A wave's DMAs land in issue order: before group gi all but the newest WROUNDS (group gi + 1's)
  // must have landed.  The halo is single-buffered.
  const int nchunk = g.Cin / 32;
  // (named arrays, never a pointer table: the wait-count pass must see which LDS object a pending DMA targets)
#define RING_(i_) ((i_) % 3 == 0 ? wb0 : ((i_) % 3 == 1 ? wb1 : wb2))
  issue_w(0, 0, wb0);
  issue_w(0, 1, wb1);
  for (int c = 0; c < nchunk; ++c) {
    const bool last = c + 1 == nchunk;
    __builtin_amdgcn_s_barrier();          // every wave is done with the halo of the last chunk
    issue_halo(c);
#pragma unroll
    for (int gi = 0; gi < 9; ++gi) {
      if (DRAM_BF16_ABL == 1 || DRAM_BF16_ABL == 2) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (DRAM_BF16_ABL == 7) { if (gi == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * WROUNDS) : "memory"); }
      else if (gi == 0 || (gi == 8 && last)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WROUNDS) : "memory");
      if (DRAM_BF16_ABL != 6) __builtin_amdgcn_s_barrier();        // group gi's weights (gi == 0: and the halo) are in LDS for every wave;
                                           // slot (gi + 2) % 3, read during group gi - 1, is free
      // (group 0 refills at its END instead: the wait-count pass, which does not see the hand-placed wait, guards
      // the first halo read with its own vmcnt(0) and would wait for a refill issued in front of it)
      if (gi != 0) {
        if (gi + 2 < 9) issue_w(c, gi + 2, RING_(gi + 2));
        else if (!last) issue_w(c + 1, gi + 2 - 9, RING_(gi + 2));
      }
      __builtin_amdgcn_sched_barrier(0);   // the refill is issued HERE, two groups ahead, not sunk behind the MFMAs
      const unsigned char* wl = RING_(gi);
      const int tapo = ((gi / 3) * 100 + (gi % 3) * 10) * 64;
      // six (kx, k16) steps; the operand fragments of step s + 1 are read while the MFMAs of step s run
      bf16x8 af[2][2], bfr[2][NB];
      auto frag = [&](int st, int buf) __attribute__((always_inline)) {
        const int kx = st >> 1, j = st & 1;
        if (DRAM_BF16_ABL == 3) {
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int e = 0; e < 8; ++e) af[buf][mi][e] = (__bf16)(float)(lane + st);
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 8; ++e) bfr[buf][nb][e] = (__bf16)(float)(lane - st);
          return;
        }
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
          af[buf][mi] = *reinterpret_cast<const bf16x8*>(halo + abase[mi] + tapo + kx * 64 + (((2 * j + lh) ^ akey[kx]) << 4));
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
          bfr[buf][nb] = *reinterpret_cast<const bf16x8*>(wl + ((kx * 32 * NB + nb * 32 + li) << 6) + (((2 * j + lh) ^ bkey) << 4));
      };
      frag(0, 0);
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        if (st + 1 < 6) frag(st + 1, (st + 1) & 1);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int nb = 0; nb < NB; ++nb)
            if (DRAM_BF16_ABL == 4) acc[mi][nb][0] += (float)af[st & 1][mi][0] * (float)bfr[st & 1][nb][0];
            else acc[mi][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[st & 1][mi], bfr[st & 1][nb], acc[mi][nb], 0, 0, 0);
        if (st + 1 < 6) __builtin_amdgcn_sched_group_barrier(0x100, 2 + NB, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 2 * NB, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (gi == 0) {
        issue_w(c, 2, wb2);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }
#undef RING_

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
          if (DRAM_BF16_ABL != 5 || v == 12345.f) y[o + nb * 32] = h;
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
    float* red = reinterpret_cast<float*>(halo);   // [NW waves][2][32 NB]
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
      for (int wv = 0; wv < NW; ++wv) v += red[(wv * 2 + which) * 32 * NB + cc];
      stats[((long)(L / g.n_tiles) * 2 + which) * g.Cout + co0 + cc] = v;
    }
  }
#endif
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

  // Per-lane DMA geometry, fixed for the whole kernel: the halo voxel (hz, hy, hx) / tile voxel (vz, vy, vx) a granule
  // belongs to and its byte offset relative to the tile's first z plane (32-bit; negative before that plane's first
  // row is fine: such lanes are out of the volume and never used).  Per tile only the plane base (scalar, in the
  // buffer descriptor) and the range tests remain -- the first version re-derived everything per tile with 64-bit
  // arithmetic (~45 VALU instructions per 1-KB piece).
  int hco[5], hrel[5], vco[4], vrel[4];
#pragma unroll
  for (int i = 0; i < 5; ++i) {              // halo: 2560 granules / 512 threads
    const int p = i * 512 + tid;
    const int hv = p >> 2, ps = p & 3;
    const int hz = hv / 100, hy = (hv / 10) % 10, hx = hv % 10;
    hco[i] = hv < 600 ? (hz << 16) | (hy << 8) | hx : -1;
    hrel[i] = ((((hz - 1) * g.H + (hy - 1)) * g.W + (hx - 1)) * g.d * g.Cin + cib * 32 + (ps ^ ((hx >> 1) & 3)) * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {              // dy tile: 256 voxels x 8 slots (64 co); slot ^= 4 * ((v >> 1) & 1)
    const int q = i * 512 + tid;
    const int v = q >> 3, ps = q & 7;
    const int ch = cob * 64 + (ps ^ (((v >> 1) & 1) << 2)) * 8;
    vco[i] = ch < g.Cout ? ((v >> 6) << 16) | (((v >> 3) & 7) << 8) | (v & 7) : -1;
    vrel[i] = ((((v >> 6) * g.H + ((v >> 3) & 7)) * g.W + (v & 7)) * g.d * g.Cout + ch) * 2;
  }
  const long xbytes = (long)g.B * g.D * g.H * g.W * g.Cin * 2, dbytes = (long)g.B * g.D * g.H * g.W * g.Cout * 2;
  auto issue_tile = [&](int t, unsigned char* hb, unsigned char* db) __attribute__((always_inline)) {
    if (DRAM_BF16_ABL == 11) return;
    int b, rz, ry, rx, lz0, ly0, lx0;
    decode_tile(t, cg, b, rz, ry, rx, lz0, ly0, lx0);
    // descriptor bases at the tile's origin voxel (rz + d lz0, ry + d ly0, rx + d lx0): always inside the tensor's
    // address range for tiles that exist; lanes whose voxel is outside the volume get the out-of-range offset
    const int oz = rz + g.d * lz0, oy = ry + g.d * ly0, ox = rx + g.d * lx0;
    const long org = (((long)b * g.D + oz) * g.H + oy) * g.W + ox;
    // (the halo starts one lattice step BEFORE the origin: offsets are taken from a base moved back by the largest
    // negative reach, (H W + W + 1) d voxels, clamped into the tensor, and corrected per lane by the same amount)
    const long back = ((long)g.H * g.W + g.W + 1) * g.d;
    const long hb0 = org - back < 0 ? 0 : org - back;
    const int hshift = (int)((org - hb0) * g.Cin * 2);
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(x, hb0 * g.Cin * 2, xbytes), rd = make_rsrc(dy, org * g.Cout * 2, dbytes);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int hz = hco[i] >> 16, hy = (hco[i] >> 8) & 255, hx = hco[i] & 255;
      const int gz = oz + g.d * (hz - 1), gy = oy + g.d * (hy - 1), gx = ox + g.d * (hx - 1);
      const bool ok = (hco[i] >= 0) & (gz >= 0) & (gz < g.D) & (gy >= 0) & (gy < g.H) & (gx >= 0) & (gx < g.W);
      BUFLDS16(rh, ok ? (unsigned)(hrel[i] + hshift) : 0xffffffffu, hb + i * 8192 + wave * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vz = vco[i] >> 16, vy = (vco[i] >> 8) & 255, vx = vco[i] & 255;
      const bool ok = (vco[i] >= 0) & (oz + g.d * vz < g.D) & (oy + g.d * vy < g.H) & (ox + g.d * vx < g.W);
      BUFLDS16(rd, ok ? (unsigned)vrel[i] : 0xffffffffu, db + i * 8192 + wave * 1024);
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
    // The operand fragments of k16 step ks + 1 (one dy fragment, seven x fragments: 16 transposed reads) are issued
    // while the seven MFMAs of step ks run.  Left to the compiler every MFMA waited for reads issued ONE MFMA
    // earlier (32 cycles of cover for >100 cycles of LDS latency): 2.7x the matrix-pipe time per tile.  All seven
    // accumulators are always fed (a wave with six taps feeds its seventh from tap 0 and never stores it).
    bf16x8 afr[2], bq[2][7];
    auto frag = [&](int ks, int bufi) __attribute__((always_inline)) {
      const int zt = ks >> 2, yp = ks & 3;
      const unsigned char* da = db + (zt * 64 + yp * 16) * 128;            // voxel (zt, 2 yp, 0)
      const unsigned char* ha = hb + (zt * 100 + yp * 20) * 64;            // halo voxel (zt, 2 yp, 0)
      if (DRAM_BF16_ABL == 12) {
        for (int e = 0; e < 8; ++e) afr[bufi][e] = (__bf16)(float)(lane + ks);
#pragma unroll
        for (int i = 0; i < 7; ++i)
          for (int e = 0; e < 8; ++e) bq[bufi][i][e] = (__bf16)(float)(lane - i);
        return;
      }
      afr[bufi] = tr8(da, aoff[0], aoff[1]);
#pragma unroll
      for (int i = 0; i < 7; ++i) bq[bufi][i] = tr8(ha, boff[i][0], boff[i][1]);
    };
    frag(0, 0);
#pragma unroll 2
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 1 < 16) frag(ks + 1, (ks + 1) & 1);
#pragma unroll
      for (int i = 0; i < 7; ++i)
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks & 1], bq[ks & 1][i], acc[i], 0, 0, 0);
      if (ks + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 7, 0);
      __builtin_amdgcn_sched_barrier(0);
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

__global__ __launch_bounds__(512) void wgrad3b_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
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

  // Per-lane DMA geometry, fixed for the whole kernel: the halo voxel (hz, hy, hx) / tile voxel (vz, vy, vx) a granule
  // belongs to and its byte offset relative to the tile's first z plane (32-bit; negative before that plane's first
  // row is fine: such lanes are out of the volume and never used).  Per tile only the plane base (scalar, in the
  // buffer descriptor) and the range tests remain -- the first version re-derived everything per tile with 64-bit
  // arithmetic (~45 VALU instructions per 1-KB piece).
  int hco[5], hrel[5], vco[4], vrel[4];
#pragma unroll
  for (int i = 0; i < 5; ++i) {              // halo: 2560 granules / 512 threads
    const int p = i * 512 + tid;
    const int hv = p >> 2, ps = p & 3;
    const int hz = hv / 100, hy = (hv / 10) % 10, hx = hv % 10;
    hco[i] = hv < 600 ? (hz << 16) | (hy << 8) | hx : -1;
    hrel[i] = ((((hz - 1) * g.H + (hy - 1)) * g.W + (hx - 1)) * g.d * g.Cin + cib * 32 + (ps ^ ((hx >> 1) & 3)) * 8) * 2;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {              // dy tile: 256 voxels x 8 slots (64 co); slot ^= 4 * ((v >> 1) & 1)
    const int q = i * 512 + tid;
    const int v = q >> 3, ps = q & 7;
    const int ch = cob * 64 + (ps ^ (((v >> 1) & 1) << 2)) * 8;
    vco[i] = ch < g.Cout ? ((v >> 6) << 16) | (((v >> 3) & 7) << 8) | (v & 7) : -1;
    vrel[i] = ((((v >> 6) * g.H + ((v >> 3) & 7)) * g.W + (v & 7)) * g.d * g.Cout + ch) * 2;
  }
  const long xbytes = (long)g.B * g.D * g.H * g.W * g.Cin * 2, dbytes = (long)g.B * g.D * g.H * g.W * g.Cout * 2;
  auto issue_tile = [&](int t, unsigned char* hb, unsigned char* db) __attribute__((always_inline)) {
    if (DRAM_BF16_ABL == 11) return;
    int b, rz, ry, rx, lz0, ly0, lx0;
    decode_tile(t, cg, b, rz, ry, rx, lz0, ly0, lx0);
    // descriptor bases at the tile's origin voxel (rz + d lz0, ry + d ly0, rx + d lx0): always inside the tensor's
    // address range for tiles that exist; lanes whose voxel is outside the volume get the out-of-range offset
    const int oz = rz + g.d * lz0, oy = ry + g.d * ly0, ox = rx + g.d * lx0;
    const long org = (((long)b * g.D + oz) * g.H + oy) * g.W + ox;
    // (the halo starts one lattice step BEFORE the origin: offsets are taken from a base moved back by the largest
    // negative reach, (H W + W + 1) d voxels, clamped into the tensor, and corrected per lane by the same amount)
    const long back = ((long)g.H * g.W + g.W + 1) * g.d;
    const long hb0 = org - back < 0 ? 0 : org - back;
    const int hshift = (int)((org - hb0) * g.Cin * 2);
    const __amdgpu_buffer_rsrc_t rh = make_rsrc(x, hb0 * g.Cin * 2, xbytes), rd = make_rsrc(dy, org * g.Cout * 2, dbytes);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const int hz = hco[i] >> 16, hy = (hco[i] >> 8) & 255, hx = hco[i] & 255;
      const int gz = oz + g.d * (hz - 1), gy = oy + g.d * (hy - 1), gx = ox + g.d * (hx - 1);
      const bool ok = (hco[i] >= 0) & (gz >= 0) & (gz < g.D) & (gy >= 0) & (gy < g.H) & (gx >= 0) & (gx < g.W);
      BUFLDS16(rh, ok ? (unsigned)(hrel[i] + hshift) : 0xffffffffu, hb + i * 8192 + wave * 1024);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int vz = vco[i] >> 16, vy = (vco[i] >> 8) & 255, vx = vco[i] & 255;
      const bool ok = (vco[i] >= 0) & (oz + g.d * vz < g.D) & (oy + g.d * vy < g.H) & (ox + g.d * vx < g.W);
      BUFLDS16(rd, ok ? (unsigned)vrel[i] : 0xffffffffu, db + i * 8192 + wave * 1024);
    }
  };

  // this wave: BOTH co halves of the block's 64, taps wave, wave + 8, wave + 16 (+ 24 for waves 0-2).  The first form
  // (wgrad3_bf16_kernel: one co half, seven taps per wave) reads 16 operand pieces per 7 MFMAs -- 128 transposed reads
  // of 512 B per k16 step and workgroup, 512 LDS cycles against 448 matrix cycles: LDS-bound.  Here the dy fragments
  // serve both halves' accumulators of every tap: 86 reads (344 cycles) per step for the same 54 MFMAs.
  const int ntap = wave < 3 ? 4 : 3;
  const bool has4 = wave < 3;                    // wave-uniform
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][c][e] = 0.f;

  // transposed-read addresses: 16-lane group g4 = lane >> 4 (bit 0: column half, bit 1: k half), lane 4 q + p of a
  // group addresses row q, the 8-byte piece p of the group's 16 columns
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p = lane & 3, h = g4 >> 1;
  const int cslot = (g4 & 1) * 2 + (p >> 1), cbyte = (p & 1) * 8;
  // A (dy): voxel (zt, 2 yp + h, 4 s + q) of the tile -> byte v * 128 + ((4 cb + cslot) ^ 4 ((v >> 1) & 1)) * 16 + cbyte;
  // with x = 4 s + q: (v >> 1) & 1 = (q >> 1) & 1 for both s
  int aoff[2][2];
#pragma unroll
  for (int c = 0; c < 2; ++c)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int v = h * 8 + 4 * s + q;
      aoff[c][s] = v * 128 + (((4 * c + cslot) ^ (((v >> 1) & 1) << 2)) << 4) + cbyte;
    }
  // B (x halo): voxel (zt + kz, 2 yp + h + ky, 4 s + q + kx) -> byte hv * 64 + (cslot ^ ((hx >> 1) & 3)) * 16 + cbyte
  int boff[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int tap = (wave + 8 * i) < 27 ? wave + 8 * i : 0;
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

  // The tile loop is written out for the two buffer pairs: with `hb = buf ? halo1 : halo0` the wait-count pass cannot
  // tell the operand reads of one pair from the LDS-DMA writes into the other and put "s_waitcnt vmcnt(0)" between
  // the DMA issue and the first read of EVERY tile -- the next tile's load was waited for before the current tile's
  // first MFMA (no DMA: -26 % in the ablation was exactly this).  Named objects per phase answer the alias query.
  auto compute = [&](const unsigned char* hb, const unsigned char* db) __attribute__((always_inline)) {
    // The operand fragments of k16 step ks + 1 are issued while the MFMAs of step ks run (see wgrad3_bf16_kernel).
    // The fourth tap of waves 0-2 is a block of its own behind a wave-uniform branch: its reads (next step) and its two
    // MFMAs (this step) are independent of each other.
    bf16x8 afr[2][2], bq[2][4];
    auto frag = [&](int ks, int bufi) __attribute__((always_inline)) {
      const int zt = ks >> 2, yp = ks & 3;
      const unsigned char* da = db + (zt * 64 + yp * 16) * 128;            // voxel (zt, 2 yp, 0)
      const unsigned char* ha = hb + (zt * 100 + yp * 20) * 64;            // halo voxel (zt, 2 yp, 0)
      if (DRAM_BF16_ABL == 12) {           // ablation: no LDS operand reads
        for (int e = 0; e < 8; ++e) afr[bufi][0][e] = afr[bufi][1][e] = (__bf16)(float)(lane + ks);
#pragma unroll
        for (int i = 0; i < 3; ++i)
          for (int e = 0; e < 8; ++e) bq[bufi][i][e] = (__bf16)(float)(lane - i);
        return;
      }
      afr[bufi][0] = tr8(da, aoff[0][0], aoff[0][1]);
      afr[bufi][1] = tr8(da, aoff[1][0], aoff[1][1]);
#pragma unroll
      for (int i = 0; i < 3; ++i) bq[bufi][i] = tr8(ha, boff[i][0], boff[i][1]);
    };
    auto frag4 = [&](int ks, int bufi) __attribute__((always_inline)) {
      const int zt = ks >> 2, yp = ks & 3;
      if (DRAM_BF16_ABL == 12) { for (int e = 0; e < 8; ++e) bq[bufi][3][e] = (__bf16)(float)(lane - 3); return; }
      bq[bufi][3] = tr8(hb + (zt * 100 + yp * 20) * 64, boff[3][0], boff[3][1]);
    };
    frag(0, 0);
    if (has4) frag4(0, 0);
#pragma unroll 2
    for (int ks = 0; ks < 16; ++ks) {
      if (ks + 1 < 16 && DRAM_BF16_ABL != 14) frag(ks + 1, (ks + 1) & 1);     // ablation 14: operands read once per tile
      if (DRAM_BF16_ABL == 14 && ks == 0) frag(1, 1);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (DRAM_BF16_ABL == 13) acc[i][c][0] += (float)afr[ks & 1][c][0] * (float)bq[ks & 1][i][0];   // ablation: no MFMAs
          else acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks & 1][c], bq[ks & 1][i], acc[i][c], 0, 0, 0);
      if (ks + 1 < 16) __builtin_amdgcn_sched_group_barrier(0x100, 10, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (has4) {
        if (ks + 1 < 16 && DRAM_BF16_ABL != 14) frag4(ks + 1, (ks + 1) & 1);
        if (DRAM_BF16_ABL == 14 && ks == 0) frag4(1, 1);
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (DRAM_BF16_ABL == 13) acc[3][c][0] += (float)afr[ks & 1][c][0] * (float)bq[ks & 1][3][0];
          else acc[3][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(afr[ks & 1][c], bq[ks & 1][3], acc[3][c], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if (split < g.ntile) issue_tile(split, halo0, dyt0);
  for (int t = split; t < g.ntile; t += 2 * g.nsplit) {
    __syncthreads();                                     // tile t has landed in pair 0; pair 1 is free
    const bool more = t + g.nsplit < g.ntile;
    if (more) issue_tile(t + g.nsplit, halo1, dyt1);
    compute(halo0, dyt0);
    __syncthreads();                                     // tile t + nsplit has landed in pair 1; pair 0 is free
    if (t + 2 * g.nsplit < g.ntile) issue_tile(t + 2 * g.nsplit, halo0, dyt0);
    if (more) compute(halo1, dyt1);
  }

  // slab[split][pair][tap][co 64][ci 32]
  const int li = lane & 31, lh = lane >> 5;
  float* sb = slab + ((long)split * g.npairs + pair) * 27 * 2048;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < ntap) {
      const int tap = wave + 8 * i;
#pragma unroll
      for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
          sb[(tap * 64 + c * 32 + row) * 32 + li] = acc[i][c][e];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight gradient, z-walking form (Cin % 64 == 0).  The tile kernel above loads 72 KB per 864 MFMAs (83 B per MFMA:
// at the full matrix rate that is 6 TB/s into LDS -- it ran at 0.28 of the pipe, bound by exactly that).  Here a
// workgroup owns a 64 x 64 (co, ci) block and WALKS an 8 x 8 column of the dilation lattice along z: per step ONE new
// 10 x 10 halo plane of x (12.5 KB) and ONE 8 x 8 plane of dy (8 KB) arrive by LDS-DMA into plane rings (4 / 2 slots)
// and serve 432 MFMAs (all 27 taps x 4 channel-block pairs x 4 k16 steps): 48 B per MFMA, no z halo re-reads.
// 8 waves; wave = (32-channel half ib of the ci block) x (taps t0, t0 + 4, ...) x both co halves: 14 (12)
// accumulators.  Each 32-channel half of a plane is its own [voxel][64 B] image, so the four voxel rows of a
// transposed read are 256 contiguous bytes (conflict-free without a swizzle) and tap offsets are address constants.
struct ZGeom {
  int B, D, H, W, Cin, Cout, d;
  int Ty, Tx;            // in-plane lattice tiles
  int Lz, nzs, lseg;     // lattice depth, z segments per column, planes per segment
  int ncol;              // columns = B * d^3 * Ty * Tx * nzs
  int nsplit;            // slabs
  int ci_blocks, co_blocks, npairs;
  int nblk;
};

// NCB: 32-channel co halves of the block that exist (1 for Cout = 32 -- the us3 layer: the second half would multiply
// zero rows; 7 accumulators per wave instead of 14)
template <int NCB>
__global__ __launch_bounds__(512) void wgrad3z_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           float* __restrict__ slab, const ZGeom g) {
  __shared__ __attribute__((aligned(1024))) unsigned char xr0[16384];   // x plane ring: [2 ib][100 voxels][64 B], padded
  __shared__ __attribute__((aligned(1024))) unsigned char xr1[16384];
  __shared__ __attribute__((aligned(1024))) unsigned char xr2[16384];
  __shared__ __attribute__((aligned(1024))) unsigned char xr3[16384];
  __shared__ __attribute__((aligned(1024))) unsigned char dr0[8192];    // dy plane ring: [2 cb][64 voxels][64 B]
  __shared__ __attribute__((aligned(1024))) unsigned char dr1[8192];
  // per-lane DMA offsets of the current column (x: two granules, dy: one): kept in LDS, not in registers -- the kernel
  // sits at the 256-VGPR limit (224 accumulator registers), every spilled value is a scratch load that counts on
  // vmcnt like the LDS-DMA, and the reload's "s_waitcnt vmcnt(0)" drained the look-ahead plane in every step
  __shared__ unsigned offtab[3 * 512];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, g.nblk);
  const int pair = L % g.npairs, split = L / g.npairs;
  const int cib = pair % g.ci_blocks, cob = pair / g.ci_blocks;
  const long plane_x = (long)g.H * g.W * g.Cin * 2, plane_dy = (long)g.H * g.W * g.Cout * 2;     // bytes per z plane

  const int ib = wave & 1, t0 = wave >> 1;
  const int ntap = t0 < 3 ? 7 : 6;
  f32x16 acc[7][NCB];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][cb][e] = 0.f;

  // transposed-read lane addresses (see wgrad3_bf16_kernel): row q of the group's block, 8-byte piece p
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3, h = g4 >> 1;
  const int cpiece = ((g4 & 1) * 2 + (p4 >> 1)) * 16 + (p4 & 1) * 8;
  int a_lane[2], b_lane[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    a_lane[s] = (h * 8 + 4 * s + q) * 64 + cpiece;                       // dy voxel (2 ks + h, 4 s + q), + cb * 4096
    b_lane[s] = ib * 6400 + (h * 10 + 4 * s + q) * 64 + cpiece;          // x halo voxel (2 ks + h + ky, 4 s + q + kx)
  }
  // Taps of tap group t0 (accumulator i): i < 6 -> z tap kz = i >> 1, in-plane tap j = 2 t0 + (i & 1) (of 0..7);
  // i == 6 -> (kz = t0, j = 8) for t0 < 3.  The z tap of an accumulator is a COMPILE-TIME value: the ring slot an x
  // fragment is read from is then a named array.  With kz a per-wave value (taps t0, t0 + 4, ...) the slot was a
  // run-time select that folded into address arithmetic, the reads lost their underlying object, and the wait-count
  // pass put "s_waitcnt vmcnt(0)" in front of them -- every step waited for the look-ahead plane it had just issued.
  int tapoff[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int j = i < 6 ? 2 * t0 + (i & 1) : 8;
    tapoff[i] = ((j / 3) * 10 + j % 3) * 64;
  }
  auto tr8 = [&](const unsigned char* base, int o0, int o1) __attribute__((always_inline)) {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o0));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + o1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7));
  };
#define XRING_(i_) (((i_) & 3) == 0 ? xr0 : (((i_) & 3) == 1 ? xr1 : (((i_) & 3) == 2 ? xr2 : xr3)))
#define DRING_(i_) (((i_) & 1) == 0 ? dr0 : dr1)

  for (int col = split; col < g.ncol; col += g.nsplit) {
    int t = col;
    const int seg = t % g.nzs; t /= g.nzs;
    const int txi = t % g.Tx; t /= g.Tx;
    const int tyi = t % g.Ty; t /= g.Ty;
    const int rx = t % g.d; t /= g.d;
    const int ry = t % g.d; t /= g.d;
    const int rz = t % g.d;
    const int b = t / g.d;
    const int lzr = (g.D - rz + g.d - 1) / g.d;                          // lattice planes of this residue class
    const int z0 = seg * g.lseg, z1 = (z0 + g.lseg < lzr) ? z0 + g.lseg : lzr;
    if (z0 >= z1) continue;                                              // (uniform)
    // per-lane DMA sources inside a plane: x granule pp -> (ib, halo voxel, slot); dy granule tid -> (cb, voxel, slot)
    __builtin_amdgcn_s_barrier();                        // every wave is done with the rings of the last column
    {
    unsigned xoff[2], doff;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int pp = r * 512 + tid;
      const int ibb = pp / 400, rr = pp % 400, hv = rr >> 2, sl = rr & 3;
      const int gy = ry + g.d * (tyi * 8 + hv / 10 - 1), gx = rx + g.d * (txi * 8 + hv % 10 - 1);
      const bool ok = (pp < 800) & (gy >= 0) & (gy < g.H) & (gx >= 0) & (gx < g.W);
      xoff[r] = ok ? (unsigned)((((long)gy * g.W + gx) * g.Cin + cib * 64 + ibb * 32 + sl * 8) * 2) : 0xffffffffu;
    }
    {
      const int cbb = tid >> 8, rr = tid & 255, v = rr >> 2, sl = rr & 3;
      const int gy = ry + g.d * (tyi * 8 + (v >> 3)), gx = rx + g.d * (txi * 8 + (v & 7));
      const int ch = cob * 64 + cbb * 32 + sl * 8;
      const bool ok = (gy < g.H) & (gx < g.W) & (ch < g.Cout);
      doff = ok ? (unsigned)((((long)gy * g.W + gx) * g.Cout + ch) * 2) : 0xffffffffu;
    }
    offtab[tid] = xoff[0];
    offtab[512 + tid] = xoff[1];
    offtab[1024 + tid] = doff;
    }
    // plane zl of the lattice (any integer): a descriptor over that plane alone, EMPTY (-> zeros) outside the volume
    auto issue_x = [&](int zl, unsigned char* dst) __attribute__((always_inline)) {
      const int gz = rz + g.d * zl;
      const bool in = (zl >= 0) & (gz < g.D);
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(x)) + ((long)b * g.D + (in ? gz : 0)) * plane_x, 0,
          in ? (int)plane_x : 0, BUF_FLAGS);
      BUFLDS16(rs, offtab[tid], dst + wave * 1024);
      BUFLDS16(rs, offtab[512 + tid], dst + 8192 + wave * 1024);
    };
    auto issue_d = [&](int zl, unsigned char* dst) __attribute__((always_inline)) {
      const int gz = rz + g.d * zl;
      const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<char*>(reinterpret_cast<const char*>(dy)) + ((long)b * g.D + gz) * plane_dy, 0, (int)plane_dy, BUF_FLAGS);
      BUFLDS16(rs, offtab[1024 + tid], dst + wave * 1024);
    };
    // ring slots are indexed by (plane - z0 + 1) for x and (plane - z0) for dy: the same named slots for every column
    issue_x(z0 - 1, xr0);
    issue_x(z0, xr1);
    issue_x(z0 + 1, xr2);
    issue_d(z0, dr0);
    for (int zz = 0; zz < z1 - z0; zz += 4) {            // four steps per trip: ring slots are compile-time names
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int z = zz + u;                            // relative plane
        if (z < z1 - z0) {                               // (uniform)
          WAIT_VM0();
          __builtin_amdgcn_s_barrier();                  // planes z - 1 .. z + 1 of x and plane z of dy are in LDS
          issue_x(z0 + z + 2, XRING_(u + 3));            // relative plane z + 2 -> slot (z + 3) & 3 = (u + 3) & 3
          if (z + 1 < z1 - z0) issue_d(z0 + z + 1, DRING_(u + 1));
          __builtin_amdgcn_sched_barrier(0);
          const unsigned char* dl = DRING_(u);
          // x fragments are read TWO taps (four MFMAs, 128 cycles: the LDS latency) ahead of their use; all seven
          // tap slots are always fed (a wave with six taps feeds the seventh from tap 0 and never stores it)
#pragma unroll 1
          for (int ks = 0; ks < 4; ++ks) {
            const bf16x8 a0 = tr8(dl + ks * 1024, a_lane[0], a_lane[1]);
            bf16x8 a1 = a0;
            if (NCB == 2) a1 = tr8(dl + 4096 + ks * 1024, a_lane[0], a_lane[1]);
            bf16x8 bq[7];
            auto bfrag = [&](int i) __attribute__((always_inline)) {
              // x plane z + kz - 1 lives in slot (u + kz) & 3 (named arrays: see RING_ in conv3_bf16_kernel)
              if (i < 6) {
                bq[i] = tr8(XRING_(u + (i >> 1)) + ks * 1280 + tapoff[i], b_lane[0], b_lane[1]);
              } else {                                   // the seventh tap: z tap t0 (wave-uniform branch, named slots)
                if (t0 == 0) bq[i] = tr8(XRING_(u) + ks * 1280 + tapoff[i], b_lane[0], b_lane[1]);
                else if (t0 == 1) bq[i] = tr8(XRING_(u + 1) + ks * 1280 + tapoff[i], b_lane[0], b_lane[1]);
                else bq[i] = tr8(XRING_(u + 2) + ks * 1280 + tapoff[i], b_lane[0], b_lane[1]);
              }
            };
#ifndef DRAM_ZW_AHEAD
#define DRAM_ZW_AHEAD 1          // x fragments read this many taps ahead (2: four more live registers -> spills)
#endif
#pragma unroll
            for (int i = 0; i < DRAM_ZW_AHEAD; ++i) bfrag(i);
#pragma unroll
            for (int i = 0; i < 7; ++i) {
              if (i + DRAM_ZW_AHEAD < 7) bfrag(i + DRAM_ZW_AHEAD);
              acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bq[i], acc[i][0], 0, 0, 0);
              if (NCB == 2) acc[i][NCB - 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bq[i], acc[i][NCB - 1], 0, 0, 0);
              if (i + DRAM_ZW_AHEAD < 7) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x008, NCB, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
    WAIT_VM0();     // the look-ahead plane of the last step: drained before reuse
  }
#undef XRING_
#undef DRING_

  // slab[split][pair][tap][co 64][ci 64]
  const int li = lane & 31, lh = lane >> 5;
  float* sb = slab + ((long)split * g.npairs + pair) * 27 * 4096;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    if (i < ntap) {
      const int tap = i < 6 ? (i >> 1) * 9 + 2 * t0 + (i & 1) : t0 * 9 + 8;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
          sb[(tap * 64 + cb * 32 + row) * 64 + ib * 32 + li] = acc[i][cb][e];
        }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// 1x1x1 convolutions (the Bottleneck blocks of ResNet-50, med3d.py:147-162) as plain bf16 GEMMs over the flat
// voxel index.  All of them are HBM-bound (256->64 at 32x64x64: 84 MB moved for 4.3 GFLOP), so the tiles are simple:
//   gemm1_bf16_kernel<NB, EPI>   y[m][n] = sum_k x[m][k] w[n][k]: 256 rows x 32 NB columns per workgroup (4 waves),
//                                64-channel K chunks, A and W tiles double-buffered by LDS-DMA, 128-B rows swizzled
//                                in 16-B slots by (row >> 1) & 7; epilogues as in conv3_bf16_kernel.
//   wgrad1_bf16_kernel           dW[co][ci] = sum_m dy[m][co] x[m][ci]: a 64 x 64 (co, ci) block per workgroup, one
//                                32 x 32 accumulator per wave, 256-voxel tiles of dy and x (two [voxel][64 B] images
//                                each) double-buffered, transposed LDS reads; voxel splits -> slabs -> ordered reduce.
struct G1Geom {
  long M;              // voxels
  int Cin, Cout;
  int m_tiles, n_tiles, nblk;
};

template <int NB, int EPI>
__global__ __launch_bounds__(256, 2) void gemm1_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ w,
                                                            const bf16_t* __restrict__ add,
                                                            const bf16_t* __restrict__ gate, bf16_t* __restrict__ y,
                                                            float* __restrict__ stats, const G1Geom g) {
#if defined(__HIP_DEVICE_COMPILE__)
  constexpr int WR = (32 * NB * 8 + 255) / 256;                  // weight DMA rounds per chunk
  __shared__ __attribute__((aligned(1024))) unsigned char a0[32768];
  __shared__ __attribute__((aligned(1024))) unsigned char a1[32768];
  __shared__ __attribute__((aligned(1024))) unsigned char w0[WR * 4096];
  __shared__ __attribute__((aligned(1024))) unsigned char w1[WR * 4096];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int L = xcd_remap(blockIdx.x, g.nblk);
  const int nt = L % g.n_tiles, mt = L / g.n_tiles;
  const long m0 = (long)mt * 256;
  const int co0 = nt * 32 * NB;
  const long xbytes = g.M * g.Cin * 2, wbytes = (long)g.Cout * g.Cin * 2;
  unsigned aoff[8], woff[WR];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int p = i * 256 + tid, r = p >> 3, ps = p & 7;
    aoff[i] = (m0 + r < g.M) ? (unsigned)(((m0 + r) * g.Cin + (ps ^ ((r >> 1) & 7)) * 8) * 2) : 0xffffffffu;
  }
#pragma unroll
  for (int i = 0; i < WR; ++i) {
    const int p = i * 256 + tid, r = p >> 3, ps = p & 7;
    woff[i] = r < 32 * NB ? (unsigned)((((long)co0 + r) * g.Cin + (ps ^ ((r >> 1) & 7)) * 8) * 2) : 0xffffffffu;
  }
  auto issue = [&](int c, unsigned char* ab, unsigned char* wb) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t ra = make_rsrc(x, (long)c * 128, xbytes), rw = make_rsrc(w, (long)c * 128, wbytes);
#pragma unroll
    for (int i = 0; i < 8; ++i) BUFLDS16(ra, aoff[i], ab + i * 4096 + wave * 1024);
#pragma unroll
    for (int i = 0; i < WR; ++i) BUFLDS16(rw, woff[i], wb + i * 4096 + wave * 1024);
  };
  f32x16 acc[2][NB];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[mi][nb][e] = 0.f;
  const int arow = wave * 64 + li;
  const int akey0 = (arow >> 1) & 7, akey1 = ((arow + 32) >> 1) & 7, bkey = (li >> 1) & 7;
  const int nchunk = g.Cin / 64;
  issue(0, a0, w0);
  for (int c = 0; c < nchunk; ++c) {
    WAIT_VM0();
    __builtin_amdgcn_s_barrier();                       // chunk c is in LDS for every wave; the other stage is free
    if (c + 1 < nchunk) {
      if (c & 1) issue(c + 1, a0, w0);
      else issue(c + 1, a1, w1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned char* al = (c & 1) ? a1 : a0;
    const unsigned char* wl = (c & 1) ? w1 : w0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int sl = 2 * j + lh;
      const bf16x8 f0 = *reinterpret_cast<const bf16x8*>(al + arow * 128 + ((sl ^ akey0) << 4));
      const bf16x8 f1 = *reinterpret_cast<const bf16x8*>(al + (arow + 32) * 128 + ((sl ^ akey1) << 4));
#pragma unroll
      for (int nb = 0; nb < NB; ++nb) {
        const bf16x8 bf = *reinterpret_cast<const bf16x8*>(wl + (nb * 32 + li) * 128 + ((sl ^ bkey) << 4));
        acc[0][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f0, bf, acc[0][nb], 0, 0, 0);
        acc[1][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f1, bf, acc[1][nb], 0, 0, 0);
      }
    }
  }
  // Epilogue through LDS.  The 32x32 accumulator layout gives a lane ONE column: direct stores are 2-byte stores in
  // 64-B pieces (and, for the shortcut-gradient epilogue, 2-byte loads of add and gate) -- 64-192 vector-memory
  // instructions per lane for a 256 x 64 tile of an HBM-bound GEMM (1024->256 data gradient with add * (gate > 0):
  // 184 us against 26 us for the forward of the same layer).  Each wave turns its 32 x 32 NB half-tile through a
  // private LDS region (row pitch 32 NB + 4 floats: conflict-free 16-B reads) and moves 8 channels = 16 B per lane.
  constexpr int CB = NB >= 2 ? 2 : 1;             // 32-column blocks per round (64 columns at most: 2 waves share 32 KB)
  constexpr int NR = NB / CB;                     // column rounds
  constexpr int P = 32 * CB + 4;                  // floats per LDS row
  constexpr int CG = 4 * CB;                      // 8-channel groups per row
  constexpr int RPP = 64 / CG;                    // rows per pass
  __syncthreads();                                // every wave is done with the operand stages
  float* reg = reinterpret_cast<float*>(wave < 2 ? a0 : a1) + (wave & 1) * (32 * P);
  const int cg = lane % CG, rs = lane / CG;
  float s1[NR][8], s2[NR][8];
#pragma unroll
  for (int cr = 0; cr < NR; ++cr)
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[cr][j] = 0.f; s2[cr][j] = 0.f; }
  // The shortcut-gradient operands (add, gate: 16 B per lane and pass) are loaded one (column round, row half) GROUP
  // ahead: issued right before use, each of the 16 passes of a workgroup waited out one memory latency -- the whole
  // epilogue of an HBM-bound GEMM with 8 K steps (1024->256 data gradient: 55 us against a 24-us traffic floor).
  constexpr int NPS = 32 / RPP;                   // passes per group
  constexpr int NG = NR * 2;                      // groups: (column round, row half)
  uint4 pa[2][NPS], pg[2][NPS];
  auto prefetch = [&](const int gi, const int buf) __attribute__((always_inline)) {
    const int cr = gi >> 1, mi = gi & 1;
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const long m = m0 + wave * 64 + mi * 32 + ps * RPP + rs;
      const long o = m * g.Cout + co0 + cr * 32 * CB + 8 * cg;
      pa[buf][ps] = make_uint4(0u, 0u, 0u, 0u);
      pg[buf][ps] = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);       // gate absent: all ones (> 0)
      if (m < g.M) {
        pa[buf][ps] = *reinterpret_cast<const uint4*>(add + o);
        if (gate) pg[buf][ps] = *reinterpret_cast<const uint4*>(gate + o);
      }
    }
  };
  const bool has_add = EPI == 1 && add;
  if (has_add) prefetch(0, 0);
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int cr = gi >> 1, mi = gi & 1;
    if (has_add && gi + 1 < NG) prefetch(gi + 1, (gi + 1) & 1);
#pragma unroll
    for (int e = 0; e < 16; ++e)
#pragma unroll
      for (int nb = 0; nb < CB; ++nb) reg[((e & 3) + 8 * (e >> 2) + 4 * lh) * P + nb * 32 + li] = acc[mi][cr * CB + nb][e];
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps) {
      const int row = ps * RPP + rs;
      const long m = m0 + wave * 64 + mi * 32 + row;
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(reg + row * P + 8 * cg);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(reg + row * P + 8 * cg + 4);
      float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
      if (m < g.M) {
        const long o = m * g.Cout + co0 + cr * 32 * CB + 8 * cg;
        if (has_add) {
          const uint4 au = pa[gi & 1][ps], gu = pg[gi & 1][ps];
          const unsigned aw[4] = {au.x, au.y, au.z, au.w};
          const unsigned gw[4] = {gu.x, gu.y, gu.z, gu.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            if (__uint_as_float(gw[j] << 16) > 0.f) v[2 * j] += __uint_as_float(aw[j] << 16);
            if (__uint_as_float(gw[j] & 0xffff0000u) > 0.f) v[2 * j + 1] += __uint_as_float(aw[j] & 0xffff0000u);
          }
        }
        uint4 hv;
        hv.x = pack2_bf16(v[0], v[1]); hv.y = pack2_bf16(v[2], v[3]);
        hv.z = pack2_bf16(v[4], v[5]); hv.w = pack2_bf16(v[6], v[7]);
        *reinterpret_cast<uint4*>(y + o) = hv;
        if (EPI == 0) {                             // BatchNorm sums of the ROUNDED values
          const unsigned hw[4] = {hv.x, hv.y, hv.z, hv.w};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float r0 = __uint_as_float(hw[j] << 16), r1 = __uint_as_float(hw[j] & 0xffff0000u);
            s1[cr][2 * j] += r0; s2[cr][2 * j] += r0 * r0;
            s1[cr][2 * j + 1] += r1; s2[cr][2 * j + 1] += r1 * r1;
          }
        }
      }
    }
  }
  if (EPI == 0 && stats) {
    // fold the row lanes of a wave (lanes with the same 8-channel group), then the four waves through LDS
#pragma unroll
    for (int cr = 0; cr < NR; ++cr)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = CG; o < 64; o <<= 1) {
          s1[cr][j] += __shfl_xor(s1[cr][j], o, 64);
          s2[cr][j] += __shfl_xor(s2[cr][j], o, 64);
        }
      }
    __syncthreads();
    float* red = reinterpret_cast<float*>(w0);   // [4 waves][2][32 NB]  (w0: no wave's turn region)
    if (rs == 0) {
#pragma unroll
      for (int cr = 0; cr < NR; ++cr)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          red[(wave * 2 + 0) * 32 * NB + cr * 32 * CB + 8 * cg + j] = s1[cr][j];
          red[(wave * 2 + 1) * 32 * NB + cr * 32 * CB + 8 * cg + j] = s2[cr][j];
        }
    }
    __syncthreads();
    if (tid < 2 * 32 * NB) {
      const int which = tid / (32 * NB), cc = tid % (32 * NB);
      float v = 0.f;
#pragma unroll
      for (int wv = 0; wv < 4; ++wv) v += red[(wv * 2 + which) * 32 * NB + cc];
      stats[((long)mt * 2 + which) * g.Cout + co0 + cc] = v;
    }
  }
#endif
}

struct W1Geom {
  long M;
  int Cin, Cout;
  int ntile, nsplit, ci_blocks, co_blocks, npairs, nblk;
};

__global__ __launch_bounds__(256) void wgrad1_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                          float* __restrict__ slab, const W1Geom g) {
  __shared__ __attribute__((aligned(1024))) unsigned char d0[32768];   // dy tile: [2 cb][256 voxels][64 B]
  __shared__ __attribute__((aligned(1024))) unsigned char d1[32768];
  __shared__ __attribute__((aligned(1024))) unsigned char x0[32768];   // x tile:  [2 ib][256 voxels][64 B]
  __shared__ __attribute__((aligned(1024))) unsigned char x1[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, g.nblk);
  const int pair = L % g.npairs, split = L / g.npairs;
  const int cib = pair % g.ci_blocks, cob = pair / g.ci_blocks;
  const int cb = wave & 1, ib = wave >> 1;
  const long xbytes = g.M * g.Cin * 2, dbytes = g.M * g.Cout * 2;
  auto issue = [&](int t, unsigned char* db, unsigned char* xb) __attribute__((always_inline)) {
    const long m0 = (long)t * 256;
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dy, m0 * g.Cout * 2, dbytes), rx = make_rsrc(x, m0 * g.Cin * 2, xbytes);
#pragma unroll
    for (int i = 0; i < 8; ++i) {                     // granule p -> (half, voxel, slot): byte p * 16 of the image pair
      const int p = i * 256 + tid, hf = p >> 10, r = (p >> 2) & 255, sl = p & 3;
      const int cd = cob * 64 + hf * 32 + sl * 8, cx = cib * 64 + hf * 32 + sl * 8;
      const bool inr = m0 + r < g.M;
      BUFLDS16(rd, (inr && cd < g.Cout) ? (unsigned)(((long)r * g.Cout + cd) * 2) : 0xffffffffu, db + i * 4096 + wave * 1024);
      BUFLDS16(rx, (inr && cx < g.Cin) ? (unsigned)(((long)r * g.Cin + cx) * 2) : 0xffffffffu, xb + i * 4096 + wave * 1024);
    }
  };
  f32x16 acc;
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0.f;
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3, h = g4 >> 1;
  const int cpiece = ((g4 & 1) * 2 + (p4 >> 1)) * 16 + (p4 & 1) * 8;
  const int lo0 = (h * 8 + q) * 64 + cpiece, lo1 = (h * 8 + 4 + q) * 64 + cpiece;
  auto tr8 = [&](const unsigned char* base) __attribute__((always_inline)) {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + lo0));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + lo1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  // (two phases with named buffer pairs, as in wgrad3b_bf16_kernel: a selected pointer made the wait-count pass wait
  // for the NEXT tile's DMA in front of the first operand read of every tile)
  auto compute = [&](const unsigned char* db, const unsigned char* xb) __attribute__((always_inline)) {
#pragma unroll 4
    for (int ks = 0; ks < 16; ++ks)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr8(db + cb * 16384 + ks * 1024), tr8(xb + ib * 16384 + ks * 1024), acc,
                                                    0, 0, 0);
  };
  if (split < g.ntile) issue(split, d0, x0);
  for (int t = split; t < g.ntile; t += 2 * g.nsplit) {
    WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    const bool more = t + g.nsplit < g.ntile;
    if (more) issue(t + g.nsplit, d1, x1);
    __builtin_amdgcn_sched_barrier(0);
    compute(d0, x0);
    WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    if (t + 2 * g.nsplit < g.ntile) issue(t + 2 * g.nsplit, d0, x0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) compute(d1, x1);
  }
  const int li = lane & 31, lh = lane >> 5;
  float* sb = slab + ((long)split * g.npairs + pair) * 4096;
#pragma unroll
  for (int e = 0; e < 16; ++e) {
    const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
    sb[(cb * 32 + row) * 64 + ib * 32 + li] = acc[e];
  }
}

// The same weight gradient on 128 x 128 (co, ci) blocks: a wave owns 64 x 64 (four accumulators), so an A or B
// fragment read from LDS feeds two MFMAs instead of one -- 1 KB of transposed LDS reads per MFMA where the 64 x 64
// form needs 2 KB (four waves: 256 B per clock against the 128 B the LDS delivers; that form sat at ~240 TFLOP/s).
// Tiles of 128 voxels, double-buffered (2 x 64 KB); slabs in the 64 x 64-pair layout of the small kernel, so the reduce
// is shared.  Channel counts that are multiples of 64 but not of 128 run ragged blocks (zero-filled loads, skipped
// slab pairs).
struct W1bGeom {
  long M;
  int Cin, Cout;
  int ntile, nsplit, ci_blocks, co_blocks, ci2, co2, npairs, nblk;     // ci_blocks / co_blocks: 64-wide; ci2 / co2: 128-wide
};

__global__ __launch_bounds__(256) void wgrad1b_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy,
                                                           float* __restrict__ slab, const W1bGeom g) {
  __shared__ __attribute__((aligned(1024))) unsigned char d0[32768];   // dy tile: [4 cb][128 voxels][64 B]
  __shared__ __attribute__((aligned(1024))) unsigned char d1[32768];
  __shared__ __attribute__((aligned(1024))) unsigned char x0[32768];   // x tile:  [4 ib][128 voxels][64 B]
  __shared__ __attribute__((aligned(1024))) unsigned char x1[32768];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int L = xcd_remap(blockIdx.x, g.nblk);
  const int blk = L % (g.ci2 * g.co2), split = L / (g.ci2 * g.co2);
  const int cib = blk % g.ci2, cob = blk / g.ci2;                     // 128-wide block indices
  const int ch = wave & 1, ih = wave >> 1;                            // the wave's 64-channel halves (co, ci)
  const long xbytes = g.M * g.Cin * 2, dbytes = g.M * g.Cout * 2;
  auto issue = [&](int t, unsigned char* db, unsigned char* xb) __attribute__((always_inline)) {
    const long m0 = (long)t * 128;
    const __amdgpu_buffer_rsrc_t rd = make_rsrc(dy, m0 * g.Cout * 2, dbytes), rx = make_rsrc(x, m0 * g.Cin * 2, xbytes);
#pragma unroll
    for (int i = 0; i < 8; ++i) {                     // granule p -> (32-channel block, voxel, slot): byte p * 16 of the image
      const int p = i * 256 + tid, hf = p >> 9, r = (p >> 2) & 127, sl = p & 3;
      const int cd = cob * 128 + hf * 32 + sl * 8, cx = cib * 128 + hf * 32 + sl * 8;
      const bool inr = m0 + r < g.M;
      BUFLDS16(rd, (inr && cd < g.Cout) ? (unsigned)(((long)r * g.Cout + cd) * 2) : 0xffffffffu, db + i * 4096 + wave * 1024);
      BUFLDS16(rx, (inr && cx < g.Cin) ? (unsigned)(((long)r * g.Cin + cx) * 2) : 0xffffffffu, xb + i * 4096 + wave * 1024);
    }
  };
  f32x16 acc[2][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;
  const int g4 = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3, h = g4 >> 1;
  const int cpiece = ((g4 & 1) * 2 + (p4 >> 1)) * 16 + (p4 & 1) * 8;
  const int lo0 = (h * 8 + q) * 64 + cpiece, lo1 = (h * 8 + 4 + q) * 64 + cpiece;
  auto tr8 = [&](const unsigned char* base) __attribute__((always_inline)) {
    const s16x4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + lo0));
    const s16x4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(base + lo1));
    return __builtin_bit_cast(bf16x8, __builtin_shufflevector(r0, r1, 0, 1, 2, 3, 4, 5, 6, 7));
  };
  auto compute = [&](const unsigned char* db, const unsigned char* xb) __attribute__((always_inline)) {
#pragma unroll 2
    for (int ks = 0; ks < 8; ++ks) {                  // 16 voxels per step; a 32-channel block image is 128 x 64 B = 8 KB
      const bf16x8 a0 = tr8(db + (2 * ch) * 8192 + ks * 1024), a1 = tr8(db + (2 * ch + 1) * 8192 + ks * 1024);
      const bf16x8 b0 = tr8(xb + (2 * ih) * 8192 + ks * 1024), b1 = tr8(xb + (2 * ih + 1) * 8192 + ks * 1024);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
  };
  // (two phases with named buffer pairs, as in wgrad1_bf16_kernel)
  if (split < g.ntile) issue(split, d0, x0);
  for (int t = split; t < g.ntile; t += 2 * g.nsplit) {
    WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    const bool more = t + g.nsplit < g.ntile;
    if (more) issue(t + g.nsplit, d1, x1);
    __builtin_amdgcn_sched_barrier(0);
    compute(d0, x0);
    WAIT_VM0();
    __builtin_amdgcn_s_barrier();
    if (t + 2 * g.nsplit < g.ntile) issue(t + 2 * g.nsplit, d0, x0);
    __builtin_amdgcn_sched_barrier(0);
    if (more) compute(d1, x1);
  }
  // slabs: the 64 x 64 pair (co64, ci64) of this wave, laid out as the small kernel does
  const int li = lane & 31, lh = lane >> 5;
  const int co64 = cob * 2 + ch, ci64 = cib * 2 + ih;
  if (co64 < g.co_blocks && ci64 < g.ci_blocks) {
    float* sb = slab + ((long)split * g.npairs + (long)co64 * g.ci_blocks + ci64) * 4096;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int row = (e & 3) + 8 * (e >> 2) + 4 * lh;
          sb[(a * 32 + row) * 64 + b * 32 + li] = acc[a][b][e];
        }
  }
}

// dw[co][ci][tap] = sum_split slab[split][pair(co / 64, ci / 32)][tap][co % 64][ci % 32], fixed order
// dw[co][ci][tap] = sum_split slab[split][pair][tap][co % 64][ci % cw], fixed order (deterministic).
// PER_PAIR: one thread per (co, ci) -- its `taps` results are contiguous in dw and, for a fixed tap, neighbouring
// threads read neighbouring slab floats: both sides coalesced (wide layers: 512 x 512 x 27 results, few slabs).
// Otherwise one thread per result (narrow layers: few (co, ci) pairs, many slabs -- the serial sum must be short).
// Both forms were latency-bound as first written (one load in flight per thread: 54 us for a 64 x 64 x 27 reduce over
// 512 slabs, 86 us for 512 x 512 x 27 over 2): the per-pair form now carries nine taps at a time (nine independent
// loads per slab), the per-result form eight interleaved partial sums over the slabs, combined in a fixed order.
template <bool PER_PAIR>
__global__ void wgrad3_bf16_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int Cout, int Cin,
                                          int ci_blocks, int npairs, int nsplit, int cw /* ci block width: 32 | 64 */,
                                          int taps) {
  const long sstride = (long)npairs * taps * 64 * cw;
  const long n = PER_PAIR ? (long)Cout * Cin : (long)Cout * Cin * taps;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    const long r = i / Cin;
    const int co = (int)(r % Cout);
    const int pair = (co >> 6) * ci_blocks + ci / cw;
    const float* s0 = slab + (long)pair * taps * 64 * cw + (co & 63) * cw + ci % cw;
    float* o = dw + ((long)co * Cin + ci) * taps;
    if (PER_PAIR) {                       // taps == 27
      const long tstride = 64L * cw;
#pragma unroll 1
      for (int t0 = 0; t0 < 27; t0 += 9) {
        float a[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) a[t] = 0.f;
        for (int k = 0; k < nsplit; ++k) {
          const float* s = s0 + k * sstride + t0 * tstride;
#pragma unroll
          for (int t = 0; t < 9; ++t) a[t] += s[t * tstride];
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) o[t0 + t] = a[t];
      }
    } else {
      const int tap = (int)(r / Cout);
      const float* s = s0 + (long)tap * 64 * cw;
      float p[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) p[j] = 0.f;
      int k = 0;
      for (; k + 8 <= nsplit; k += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) p[j] += s[(k + j) * sstride];
      }
      for (int j = 0; k < nsplit; ++k, ++j) p[j] += s[k * sstride];
      o[tap] = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// weight packing and casts
// One tile = 32 output channels x PK_RUN / taps input channels x all taps (PK_RUN = 864 = 32 x 27 source elements per
// output channel), turned through LDS: the source rows w[co][ci0..][taps] are contiguous runs (coalesced fp32 reads),
// wf[tap][co][ci0..] leaves in runs of the tile's input channels and wb[taps-1-tap][ci][co0..] in runs of 32 output
// channels (64 bytes).  The element-per-thread form read w with 108-byte lane strides 27 times over and wrote wb two
// bytes per 2 Cout-byte stride: 0.6 ms for ResNet-18's 33 M weights, ten times the traffic's time.
constexpr int PK_RUN = 864, PK_CO = 32, PK_PITCH = PK_RUN + 2;      // pitch: 433 words (odd) -> column reads conflict-free
__host__ __device__ inline int pack_ci_per_tile(int taps) { return taps >= PK_RUN ? 1 : PK_RUN / taps; }
inline long pack_tiles(int Cout, int Cin, int taps) {
  const int ci_t = pack_ci_per_tile(taps);
  return (long)((Cout + PK_CO - 1) / PK_CO) * ((Cin + ci_t - 1) / ci_t);
}

template <int TAPS>
__device__ __forceinline__ void pack_tile_out(bf16_t* __restrict__ wf, bf16_t* __restrict__ wb, const int Cout,
                                              const int Cin, const int co0, const int ci0, const bf16_t* lds) {
  constexpr int CI = PK_RUN / TAPS, TOTAL = TAPS * PK_CO * CI;
  if (wf)
    for (int i = threadIdx.x; i < TOTAL; i += 256) {             // (tap, r, cl), cl fastest
      const int cl = i % CI, q = i / CI, r = q % PK_CO, tap = q / PK_CO;
      wf[((long)tap * Cout + co0 + r) * Cin + ci0 + cl] = lds[r * PK_PITCH + cl * TAPS + tap];
    }
  if (wb)
    for (int i = threadIdx.x; i < TOTAL; i += 256) {             // (tap, cl, r), r fastest
      const int r = i % PK_CO, q = i / PK_CO, cl = q % CI, tap = q / CI;
      wb[((long)(TAPS - 1 - tap) * Cin + ci0 + cl) * Cout + co0 + r] = lds[r * PK_PITCH + cl * TAPS + tap];
    }
}

__device__ __forceinline__ void pack_tile_bf16(const float* __restrict__ w, bf16_t* __restrict__ wf,
                                               bf16_t* __restrict__ wb, const int Cout, const int Cin, const int taps,
                                               const long tile, bf16_t* lds) {
  const int CI = pack_ci_per_tile(taps);
  const int n_ci_tiles = (Cin + CI - 1) / CI;
  const int co0 = (int)(tile / n_ci_tiles) * PK_CO, ci0 = (int)(tile % n_ci_tiles) * CI;
  const int nco = Cout - co0 < PK_CO ? Cout - co0 : PK_CO;
  const int nci = Cin - ci0 < CI ? Cin - ci0 : CI;
  const int run = nci * taps;                                    // <= PK_RUN
  if (nco == PK_CO && run == PK_RUN) {                            // full tile: one flat loop, eight loads in flight
    const float* src = w + ((long)co0 * Cin + ci0) * taps;
    const long row = (long)Cin * taps;
#pragma unroll 8
    for (int i = threadIdx.x; i < PK_CO * PK_RUN; i += 256) {
      const int r = i / PK_RUN, e = i - r * PK_RUN;
      lds[r * PK_PITCH + e] = f32_to_bf16(src[r * row + e]);
    }
  } else {
    for (int r = 0; r < nco; ++r) {
      const float* src = w + ((long)(co0 + r) * Cin + ci0) * taps;
      for (int e = threadIdx.x; e < run; e += 256) lds[r * PK_PITCH + e] = f32_to_bf16(src[e]);
    }
  }
  __syncthreads();
  if (nco == PK_CO && nci == CI && (taps == 27 || taps == 1)) {     // full tile of the two network tap counts:
    if (taps == 27) pack_tile_out<27>(wf, wb, Cout, Cin, co0, ci0, lds);   // compile-time divisors
    else pack_tile_out<1>(wf, wb, Cout, Cin, co0, ci0, lds);
    return;
  }
  if (wf) {
    const int total = taps * nco * nci;                          // (tap, r, cl), cl fastest
    for (int i = threadIdx.x; i < total; i += 256) {
      const int cl = i % nci, q = i / nci, r = q % nco, tap = q / nco;
      wf[((long)tap * Cout + co0 + r) * Cin + ci0 + cl] = lds[r * PK_PITCH + cl * taps + tap];
    }
  }
  if (wb) {
    const int total = taps * nci * nco;                          // (tap, cl, r), r fastest
    for (int i = threadIdx.x; i < total; i += 256) {
      const int r = i % nco, q = i / nco, cl = q % nci, tap = q / nci;
      wb[((long)(taps - 1 - tap) * Cin + ci0 + cl) * Cout + co0 + r] = lds[r * PK_PITCH + cl * taps + tap];
    }
  }
}

__global__ __launch_bounds__(256) void pack_weight_bf16_kernel(const float* __restrict__ w, bf16_t* __restrict__ wf,
                                                               bf16_t* __restrict__ wb, int Cout, int Cin, int taps) {
  __shared__ bf16_t lds[PK_CO * PK_PITCH];
  pack_tile_bf16(w, wf, wb, Cout, Cin, taps, blockIdx.x, lds);
}

// All bf16 weight packings of a step in ONE launch (a ResNet-50 step issued 54 launches of ~10 us): a device work list
// as in adam_multi_kernel -- DramPackRef per weight (its packed copies at fixed offsets of one flat bf16 buffer),
// DramChunkRef per tile (offset = the tile index within its weight).
__global__ __launch_bounds__(256) void pack_weight_bf16_multi_kernel(const DramPackRef* __restrict__ table,
                                                                     const DramChunkRef* __restrict__ chunks,
                                                                     bf16_t* __restrict__ flat) {
  __shared__ bf16_t lds[PK_CO * PK_PITCH];
  const DramChunkRef c = chunks[blockIdx.x];
  const DramPackRef t = table[c.tensor];
  pack_tile_bf16(t.w, t.off_f >= 0 ? flat + t.off_f : nullptr, t.off_b >= 0 ? flat + t.off_b : nullptr, t.Cout, t.Cin,
                 t.taps, c.offset, lds);
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

// ---------------------------------------------------------------------------------------------------------
// The stride-2 3x3x3 convolution (one per network: layerN.0.conv1 / the Bottleneck's conv2) as a STRIDE-1 convolution:
// x[2o + k - 1] is, per axis, the even sub-lattice at o (k = 1) or the odd one at o - 1 (k = 0) / o (k = 2).  With
// x8[o][p][c] = x[2o + p][c] (space to depth: eight parity sub-lattices side by side as 8 C channels) the convolution
// is y = conv3x3x3_stride1(x8, w3) with w3[t][co][p c] = w[k][co][c] at (p, t) = (1, -1), (0, 0), (1, 0) per axis for
// k = 0, 1, 2 and zero elsewhere (every tap with a +1 offset, and (even, -1)) -- so forward, data gradient and weight
// gradient run on the bf16 kernels above instead of the fp32 kernels between two cast passes, at 8x the products of
// which 27 / 216 are non-zero (the layer is < 1 % of the network's work).
__global__ void s2d_bf16_kernel(const uint4* __restrict__ x, uint4* __restrict__ x8, int D, int H, int W, int C8,
                                long total) {   // total = B (D/2) (H/2) (W/2) 8 C8 uint4 (8 channels each)
  const int Do = D >> 1, Ho = H >> 1, Wo = W >> 1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long r = i / C8;
    const int p = (int)(r & 7); r >>= 3;
    const int xo = (int)(r % Wo); r /= Wo;
    const int yo = (int)(r % Ho); r /= Ho;
    const int zo = (int)(r % Do);
    const long b = r / Do;
    const int z = 2 * zo + (p >> 2), y = 2 * yo + ((p >> 1) & 1), xx = 2 * xo + (p & 1);
    x8[i] = x[(((b * D + z) * H + y) * W + xx) * (long)C8 + c8];
  }
}
// dx[z][y][x][c] = dx8[z / 2][y / 2][x / 2][parity][c]  (+ add * (gate > 0): the data gradient's fused epilogue)
__global__ void d2s_bf16_kernel(const bf16_t* __restrict__ dx8, const bf16_t* __restrict__ add,
                                const bf16_t* __restrict__ gate, bf16_t* __restrict__ dx, int D, int H, int W, int C8,
                                long total) {   // total = B D H W C8
  const int Ho = H >> 1, Wo = W >> 1, Do = D >> 1;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8);
    long r = i / C8;
    const int xx = (int)(r % W); r /= W;
    const int y = (int)(r % H); r /= H;
    const int z = (int)(r % D);
    const long b = r / D;
    const int p = ((z & 1) << 2) | ((y & 1) << 1) | (xx & 1);
    const long src = (((((b * Do + (z >> 1)) * Ho + (y >> 1)) * Wo + (xx >> 1)) * 8 + p) * (long)C8 + c8) * 8;
    fvec<8> v = ldv<bf16_t, 8>(dx8, src);
    if (add) {
      const fvec<8> a = ldv<bf16_t, 8>(add, 8 * i);
      if (gate) {
        const fvec<8> g = ldv<bf16_t, 8>(gate, 8 * i);
#pragma unroll
        for (int k = 0; k < 8; ++k) v.v[k] += g.v[k] > 0.f ? a.v[k] : 0.f;
      } else {
#pragma unroll
        for (int k = 0; k < 8; ++k) v.v[k] += a.v[k];
      }
    }
    stv<bf16_t, 8>(dx, 8 * i, v);
  }
}
// per-axis map of the embedding: (parity p, offset index t = offset + 1) -> kernel tap k, or -1
__device__ __forceinline__ int s2_tap(int p, int t) { return (p == 1 && t == 0) ? 0 : ((p == 0 && t == 1) ? 1 : ((p == 1 && t == 1) ? 2 : -1)); }
// w [Cout][Cin][27] -> w3 [Cout][8 Cin][27]
__global__ void s2_embed_weight_kernel(const float* __restrict__ w, float* __restrict__ w3, int Cout, int Cin) {
  const long n = (long)Cout * 8 * Cin * 27;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int t = (int)(i % 27);
    long r = i / 27;
    const int ci = (int)(r % Cin); r /= Cin;
    const int p = (int)(r & 7);
    const long co = r >> 3;
    const int kz = s2_tap(p >> 2, t / 9), ky = s2_tap((p >> 1) & 1, (t / 3) % 3), kx = s2_tap(p & 1, t % 3);
    w3[i] = (kz < 0 || ky < 0 || kx < 0) ? 0.f : w[(co * Cin + ci) * 27 + (kz * 3 + ky) * 3 + kx];
  }
}
// dw3 [Cout][8 Cin][27] -> dw [Cout][Cin][27]
__global__ void s2_extract_wgrad_kernel(const float* __restrict__ dw3, float* __restrict__ dw, int Cout, int Cin) {
  const long n = (long)Cout * Cin * 27;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int k = (int)(i % 27);
    long r = i / 27;
    const int ci = (int)(r % Cin);
    const long co = r / Cin;
    const int kz = k / 9, ky = (k / 3) % 3, kx = k % 3;
    // k = 0 -> (p 1, t 0), k = 1 -> (0, 1), k = 2 -> (1, 1) per axis
    const int p = ((kz != 1) << 2) | ((ky != 1) << 1) | (kx != 1);
    const int t = ((kz != 0) * 3 + (ky != 0)) * 3 + (kx != 0);
    dw[i] = dw3[((co * 8 + p) * Cin + ci) * 27 + t];
  }
}

inline int ew_grid(long total) { return ew_blocks(total, 256, 8192); }   // one-shot blocks (common.h)

bool geom1_ok(const DramConvDesc* d) {       // 1x1x1, stride 1: a GEMM over the flat voxel index
  if (!d) return false;
  if (d->B < 1 || d->D < 1 || d->H < 1 || d->W < 1) return false;
  if (d->k != 1 || d->stride != 1 || d->pad != 0) return false;
  if (d->Do != d->D || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cin < 64 || d->Cout < 64 || d->Cin % 64 != 0 || d->Cout % 64 != 0) return false;
  const long long vox = (long long)d->B * d->D * d->H * d->W;
  const long long cmax = d->Cin > d->Cout ? d->Cin : d->Cout;
  return vox * cmax < (1LL << 31);
}

bool geom_ok(const DramConvDesc* d) {
  if (!d) return false;
  if (d->B < 1 || d->D < 1 || d->H < 1 || d->W < 1) return false;
  if (d->k != 3 || d->stride != 1 || d->dil < 1 || d->pad != d->dil) return false;
  if (d->Do != d->D || d->Ho != d->H || d->Wo != d->W) return false;
  if (d->Cin < 32 || d->Cout < 32 || d->Cin % 32 != 0 || d->Cout % 32 != 0) return false;
  // 32-bit byte offsets inside ONE halo window (10 z planes of the dilation lattice at most) and inside the weights;
  // tensors themselves may exceed 4 GB (64-bit bases)
  const long long cmax = d->Cin > d->Cout ? d->Cin : d->Cout;
  if (10LL * d->dil * d->H * d->W * cmax * 2 >= (1LL << 31)) return false;
  if ((long long)d->Cout * d->Cin * 27 * 2 >= (1LL << 32)) return false;
  return true;
}

// forward-sense descriptor -> lattice tiling (tz x 8 x 8 voxels per tile) of a tensor with the spatial extent (D, H, W)
void fill_tiles(const DramConvDesc* d, int& Tz, int& Ty, int& Tx, long long& ntile, int tz = 4) {
  auto tiles = [&](int n, int per) { return ((n + d->dil - 1) / d->dil + per - 1) / per; };
  Tz = tiles(d->D, tz); Ty = tiles(d->H, 8); Tx = tiles(d->W, 8);
  ntile = (long long)d->B * d->dil * d->dil * d->dil * Tz * Ty * Tx;
}

// Tile depth of the forward / data-gradient kernel: 4 z-slices (256 voxels, 4 waves, two workgroups per CU).
// Measured alternatives (tools/conv_bf16_bench.py, config 2's layers, same box): 8 waves x 512 voxels with one
// workgroup per CU (DRAM_BF16_NW=8, kept for A/B and tests): 0-5 % slower; 4 waves x 512 voxels with 16-channel
// chunks, two workgroups per CU, 100 instead of 170 bytes of LDS fill per MFMA (tried in round 3, removed): +-1 %.
// The kernel is POWER-bound on random data, not fill- or issue-bound: with all-zero activations (or weights) the
// same launches run 14-28 % faster (layer4: 1.34 -> 1.65-1.77 PFLOP/s; BENCH_ZERO=act|w) -- the chip does not hold
// its clock under dense random bf16 MFMA work, and an ablation that removes the weight DMA "gains" 40 % only because
// it feeds zeros to the matrix pipe.
int pick_nw(const DramConvDesc* d, int n_tiles) {
  (void)d; (void)n_tiles;
  if (const char* e = tune_env("DRAM_BF16_NW")) {
    const int v = atoi(e);
    if (v == 4 || v == 8) return v;
  }
  return 4;
}

// N tile: 64 columns (2 blocks); 32 when the channel count is not a multiple of 64
int pick_nb(int cout) { return cout % 64 == 0 ? 2 : 1; }

int launch_conv(const bf16_t* x, const bf16_t* w, const float* bias, const bf16_t* add, const bf16_t* gate, bf16_t* y,
                float* stats, const DramConvDesc* d, int cin, int cout, int epi, hipStream_t s) {
  CGeom g{};
  g.B = d->B; g.D = d->D; g.H = d->H; g.W = d->W; g.Cin = cin; g.Cout = cout; g.d = d->dil;
  long long ntile;
  const int nb = pick_nb(cout);
  g.n_tiles = cout / (32 * nb);
  const int nw = pick_nw(d, g.n_tiles);                                 // tile depth in z-slices
  fill_tiles(d, g.Tz, g.Ty, g.Tx, ntile, nw);
  const long long nblk = ntile * g.n_tiles;
  if (nblk >= (1LL << 31)) return DRAM_ERR_UNSUPPORTED;
  g.nblk = (int)nblk;
  const double vox = (double)d->B * d->D * d->H * d->W;
  const double flops = 2.0 * (double)ntile * 64.0 * nw * 27.0 * cin * cout;      // executed, padded tiles included
  DramProf prof(DRAM_FAM_CONV_BF16, epi * 4 + (nw == 8 ? 2 : 0) + (nb - 1), flops,
                2.0 * (vox * (cin + cout * (1.0 + (add ? 1 : 0) + (gate ? 1 : 0))) + 27.0 * cin * cout), s,
                2.0 * vox * 27.0 * cin * cout);
#define LAUNCH_(NB_, EPI_, NW_)                                                                                    \
  hipLaunchKernelGGL((conv3_bf16_kernel<NB_, EPI_, NW_>), dim3(g.nblk), dim3(64 * NW_), 0, s, x, w, bias, add, gate, y, \
                     stats, g)
#define LAUNCH_NW_(NB_, EPI_) do { if (nw == 8) LAUNCH_(NB_, EPI_, 8); else LAUNCH_(NB_, EPI_, 4); } while (0)
  if (epi == 0) { if (nb == 2) LAUNCH_NW_(2, 0); else LAUNCH_NW_(1, 0); }
  else          { if (nb == 2) LAUNCH_NW_(2, 1); else LAUNCH_NW_(1, 1); }
#undef LAUNCH_NW_
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
  // one workgroup per CU in total (147 KB of LDS: one is resident), every split at least 2 tiles deep (double
  // buffering), at most 256 slabs: a slab is 27 x 2048 floats per (pair, split) and the reduce reads them all
  // (512 -> 512 @ 2x16x32x32, ms: 1024 workgroups 0.69, 512 0.55, 256 0.56; 256 -> 256: 0.40 / 0.26 / 0.19)
  // ... and 576 -> 64 @ 2x32x64x64 (1,024 tiles, 18 pairs): 0.73 / - / 0.93 -- large volumes want the finer split
  const int wgs = tune_env("DRAM_BF16_WGRAD_WGS") ? atoi(tune_env("DRAM_BF16_WGRAD_WGS")) : (g.ntile >= 512 ? 1024 : 256);
  int ns = (wgs + g.npairs - 1) / g.npairs;
  if (ns > g.ntile / 2) ns = g.ntile / 2;
  if (ns > 256) ns = 256;
  if (ns < 1) ns = 1;
  g.nsplit = ns;
  g.nblk = g.npairs * g.nsplit;
}

// Weight-gradient form.  Measured on config 2's layers (tools/conv_bf16_bench.py, ms tile / z-walk): 64->64 @ 64x128x128
// 0.60 / 0.47, 128->64 @ 64x128x128 1.11 / 0.94, 64->32 0.58 / 0.46, 64->64 @ 32x64x64 0.157 / 0.141 -- but
// 576->64 @ 32x64x64 0.79 / 1.10 (nine ci blocks re-read every dy plane) and the 16x32x32 stages 0.076-0.70 /
// 0.12-0.76 (columns of 4-16 planes: the walk never leaves its prologue).  So: walk deep lattices with few
// channel-block pairs, tile the rest.  DRAM_BF16_WGRAD = tile | zwalk forces one (A/B, tests).
void launch_reduce(const float* slab, float* dw, int Cout, int Cin, int ci_blocks, int npairs, int nsplit, int cw, int taps,
                   hipStream_t s) {
  const long pairs = (long)Cout * Cin;
  if (taps > 1 && pairs >= 65536)
    hipLaunchKernelGGL((wgrad3_bf16_reduce_kernel<true>), dim3(ew_grid(pairs)), dim3(256), 0, s, slab, dw, Cout, Cin,
                       ci_blocks, npairs, nsplit, cw, taps);
  else
    hipLaunchKernelGGL((wgrad3_bf16_reduce_kernel<false>), dim3(ew_grid(pairs * taps)), dim3(256), 0, s, slab, dw, Cout,
                       Cin, ci_blocks, npairs, nsplit, cw, taps);
}

bool use_zwalk(const DramConvDesc* d) {
  if (d->Cin % 64 != 0) return false;
  const char* e = tune_env("DRAM_BF16_WGRAD");
  if (e && !strcmp(e, "tile")) return false;
  if (e && !strcmp(e, "zwalk")) return true;
  const int lz = (d->D + d->dil - 1) / d->dil;
  const int npairs = (d->Cin / 64) * ((d->Cout + 63) / 64);
  return lz >= 32 && npairs <= 4;
}

void plan_zwalk(const DramConvDesc* d, ZGeom& g) {
  g.B = d->B; g.D = d->D; g.H = d->H; g.W = d->W; g.Cin = d->Cin; g.Cout = d->Cout; g.d = d->dil;
  auto tiles = [&](int n, int per) { return ((n + d->dil - 1) / d->dil + per - 1) / per; };
  g.Ty = tiles(d->H, 8); g.Tx = tiles(d->W, 8);
  g.Lz = (d->D + d->dil - 1) / d->dil;
  g.ci_blocks = d->Cin / 64;
  g.co_blocks = (d->Cout + 63) / 64;
  g.npairs = g.ci_blocks * g.co_blocks;
  const long long cols = (long long)d->B * d->dil * d->dil * d->dil * g.Ty * g.Tx;
  // z segments: enough workgroups for two per CU, never shorter than 4 planes (each segment re-reads 2 halo planes)
  int nzs = 1;
  static const int zw_wgs = tune_env("DRAM_BF16_ZWALK_WGS") ? atoi(tune_env("DRAM_BF16_ZWALK_WGS")) : 512;     // A/B
  while (cols * nzs * g.npairs < zw_wgs && g.Lz / (nzs * 2) >= 4) nzs *= 2;
  g.nzs = nzs;
  g.lseg = (g.Lz + nzs - 1) / nzs;
  g.ncol = (int)(cols * nzs);
  int ns = (1024 + g.npairs - 1) / g.npairs;
  if (ns > g.ncol) ns = g.ncol;
  if (ns > 256) ns = 256;
  if (ns < 1) ns = 1;
  g.nsplit = ns;
  g.nblk = g.npairs * g.nsplit;
}

int launch_gemm1(const bf16_t* x, const bf16_t* w, const bf16_t* add, const bf16_t* gate, bf16_t* y, float* stats,
                 const DramConvDesc* d, int cin, int cout, int epi, hipStream_t s) {
  G1Geom g{};
  g.M = (long)d->B * d->D * d->H * d->W;
  g.Cin = cin; g.Cout = cout;
  g.m_tiles = (int)((g.M + 255) / 256);
  // 128-column tiles, unless they leave the chip under-filled (two workgroups per CU: 512 slots): the 16 x 32 x 32 stages
  // of ResNet-50 have 64 row tiles, so 1024->256 was 128 workgroups on 256 CUs (26 us for 42 MB: 1.6 TB/s).  64-column
  // tiles double the workgroups; the A tile they re-read comes from L2.  DRAM_BF16_GEMM1_NB=4 (DRAM_TUNING=1): A/B.
  int nb = cout % 128 == 0 ? 4 : 2;
  static const int force_nb = tune_env("DRAM_BF16_GEMM1_NB") ? atoi(tune_env("DRAM_BF16_GEMM1_NB")) : 0;
  if (nb == 4 && force_nb != 4 && (long)g.m_tiles * (cout / 128) < 512) nb = 2;
  g.n_tiles = cout / (32 * nb);
  g.nblk = g.m_tiles * g.n_tiles;
  DramProf prof(DRAM_FAM_CONV_BF16, 8 + epi * 2 + (nb == 4), 2.0 * (double)g.m_tiles * 256.0 * cin * cout,
                2.0 * ((double)g.M * (cin + cout * (1.0 + (add ? 1 : 0) + (gate ? 1 : 0))) + (double)cin * cout), s,
                2.0 * (double)g.M * cin * cout);
#define LAUNCH_(NB_, EPI_) \
  hipLaunchKernelGGL((gemm1_bf16_kernel<NB_, EPI_>), dim3(g.nblk), dim3(256), 0, s, x, w, add, gate, y, stats, g)
  if (epi == 0) { if (nb == 4) LAUNCH_(4, 0); else LAUNCH_(2, 0); }
  else          { if (nb == 4) LAUNCH_(4, 1); else LAUNCH_(2, 1); }
#undef LAUNCH_
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

bool wgrad1_big(const DramConvDesc* d) {      // 128 x 128 blocks: from 128 channels on both sides (A/B: DRAM_BF16_WGRAD1=small)
  static const bool small = tune_env("DRAM_BF16_WGRAD1") && !strcmp(tune_env("DRAM_BF16_WGRAD1"), "small");
  return !small && d->Cin >= 128 && d->Cout >= 128;
}
void plan_wgrad1b(const DramConvDesc* d, W1bGeom& g) {
  g.M = (long)d->B * d->D * d->H * d->W;
  g.Cin = d->Cin; g.Cout = d->Cout;
  g.ntile = (int)((g.M + 127) / 128);
  g.ci_blocks = d->Cin / 64;
  g.co_blocks = d->Cout / 64;
  g.ci2 = (d->Cin + 127) / 128;
  g.co2 = (d->Cout + 127) / 128;
  g.npairs = g.ci_blocks * g.co_blocks;
  const int nb = g.ci2 * g.co2;
  int ns = (512 + nb - 1) / nb;                  // ~512 workgroups (one per CU and a half: 128 KB of LDS each)
  if (ns > g.ntile / 2) ns = g.ntile / 2;
  if (ns > 256) ns = 256;
  if (ns < 1) ns = 1;
  g.nsplit = ns;
  g.nblk = nb * ns;
}
void plan_wgrad1(const DramConvDesc* d, W1Geom& g) {
  g.M = (long)d->B * d->D * d->H * d->W;
  g.Cin = d->Cin; g.Cout = d->Cout;
  g.ntile = (int)((g.M + 255) / 256);
  g.ci_blocks = d->Cin / 64;
  g.co_blocks = d->Cout / 64;
  g.npairs = g.ci_blocks * g.co_blocks;
  int ns = (1024 + g.npairs - 1) / g.npairs;
  if (ns > g.ntile / 2) ns = g.ntile / 2;
  if (ns > 256) ns = 256;
  if (ns < 1) ns = 1;
  g.nsplit = ns;
  g.nblk = g.npairs * g.nsplit;
}

}  // namespace

extern "C" int dram_conv_bf16_supported(const DramConvDesc* d) { return (geom_ok(d) || geom1_ok(d)) ? 1 : 0; }

extern "C" int dram_conv_bf16_num_stat_rows(const DramConvDesc* d) {
  if (geom1_ok(d)) return (int)(((long long)d->B * d->D * d->H * d->W + 255) / 256);
  if (!geom_ok(d)) return DRAM_ERR_UNSUPPORTED;
  int Tz, Ty, Tx;
  long long ntile;
  fill_tiles(d, Tz, Ty, Tx, ntile, pick_nw(d, d->Cout / (32 * pick_nb(d->Cout))));    // the forward launch's tiling
  return ntile < (1LL << 31) ? (int)ntile : DRAM_ERR_UNSUPPORTED;
}

extern "C" int dram_pack_conv_weight_bf16(const float* w, void* wf, void* wb, int Cout, int Cin, int taps,
                                          dram_stream_t stream) {
  if (!w || (!wf && !wb) || Cout < 1 || Cin < 1 || taps < 1) return DRAM_ERR_BAD_ARG;
  if (taps > PK_RUN) return DRAM_ERR_UNSUPPORTED;
  const long n = (long)Cout * Cin * taps;
  DramProf prof(DRAM_FAM_WEIGHT_PACK, 8, 0.0, (double)n * (4.0 + 2.0 * ((wf ? 1 : 0) + (wb ? 1 : 0))), (hipStream_t)stream);
  hipLaunchKernelGGL(pack_weight_bf16_kernel, dim3((unsigned)pack_tiles(Cout, Cin, taps)), dim3(256), 0,
                     (hipStream_t)stream, w, (bf16_t*)wf, (bf16_t*)wb, Cout, Cin, taps);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" long long dram_pack_conv_weight_bf16_tiles(int Cout, int Cin, int taps) {
  if (Cout < 1 || Cin < 1 || taps < 1) return DRAM_ERR_BAD_ARG;
  if (taps > PK_RUN) return DRAM_ERR_UNSUPPORTED;                  // (a tile holds all taps of one input channel)
  return pack_tiles(Cout, Cin, taps);
}

extern "C" int dram_pack_conv_weight_bf16_multi(const DramPackRef* table, const DramChunkRef* chunks, int nchunks,
                                                void* flat, double total_elems, dram_stream_t stream) {
  if (!table || !chunks || !flat || nchunks < 1) return DRAM_ERR_BAD_ARG;
  DramProf prof(DRAM_FAM_WEIGHT_PACK, 9, 0.0, total_elems * 8.0, (hipStream_t)stream);
  hipLaunchKernelGGL(pack_weight_bf16_multi_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, table, chunks,
                     (bf16_t*)flat);
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

extern "C" int dram_s2d_bf16(const void* x, void* x8, int B, int D, int H, int W, int C, dram_stream_t stream) {
  if (!x || !x8 || B < 1 || D < 2 || H < 2 || W < 2 || ((D | H | W) & 1) || C < 8 || (C & 7)) return DRAM_ERR_BAD_ARG;
  const long total = (long)B * D * H * W * (C >> 3);
  DramProf prof(DRAM_FAM_BN, 9, 0.0, 32.0 * (double)total, (hipStream_t)stream);
  hipLaunchKernelGGL(s2d_bf16_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const uint4*)x, (uint4*)x8,
                     D, H, W, C >> 3, total);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_d2s_bf16(const void* dx8, const void* add, const void* gate, void* dx, int B, int D, int H, int W,
                             int C, dram_stream_t stream) {
  if (!dx8 || !dx || (gate && !add) || B < 1 || D < 2 || H < 2 || W < 2 || ((D | H | W) & 1) || C < 8 || (C & 7))
    return DRAM_ERR_BAD_ARG;
  const long total = (long)B * D * H * W * (C >> 3);
  DramProf prof(DRAM_FAM_BN, 10, 0.0, 16.0 * (double)total * (2 + (add ? 1 : 0) + (gate ? 1 : 0)), (hipStream_t)stream);
  hipLaunchKernelGGL(d2s_bf16_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dx8,
                     (const bf16_t*)add, (const bf16_t*)gate, (bf16_t*)dx, D, H, W, C >> 3, total);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_s2_embed_weight(const float* w, float* w3, int Cout, int Cin, dram_stream_t stream) {
  if (!w || !w3 || Cout < 1 || Cin < 1) return DRAM_ERR_BAD_ARG;
  const long n = (long)Cout * 8 * Cin * 27;
  DramProf prof(DRAM_FAM_WEIGHT_PACK, 9, 0.0, 4.5 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(s2_embed_weight_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, w, w3, Cout, Cin);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_s2_extract_wgrad(const float* dw3, float* dw, int Cout, int Cin, dram_stream_t stream) {
  if (!dw3 || !dw || Cout < 1 || Cin < 1) return DRAM_ERR_BAD_ARG;
  const long n = (long)Cout * Cin * 27;
  DramProf prof(DRAM_FAM_WGRAD_BF16, 6, 0.0, 8.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(s2_extract_wgrad_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, dw3, dw, Cout, Cin);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_conv3d_fwd_bf16(const void* x, const void* wf, const float* bias, void* y, float* stats_partial,
                                    const DramConvDesc* d, dram_stream_t stream) {
  if (!x || !wf || !y) return DRAM_ERR_BAD_ARG;
  if (geom1_ok(d)) {
    if (bias) return DRAM_ERR_UNSUPPORTED;             // the Bottleneck 1x1x1 convolutions carry no bias
    return launch_gemm1((const bf16_t*)x, (const bf16_t*)wf, nullptr, nullptr, (bf16_t*)y, stats_partial, d, d->Cin,
                        d->Cout, 0, (hipStream_t)stream);
  }
  if (!geom_ok(d)) return DRAM_ERR_UNSUPPORTED;
  return launch_conv((const bf16_t*)x, (const bf16_t*)wf, bias, nullptr, nullptr, (bf16_t*)y, stats_partial, d, d->Cin,
                     d->Cout, 0, (hipStream_t)stream);
}

extern "C" int dram_conv3d_bwd_data_bf16(const void* dy, const void* wb, void* dx, const void* add, const void* gate,
                                         const DramConvDesc* d, dram_stream_t stream) {
  if (!dy || !wb || !dx || (gate && !add)) return DRAM_ERR_BAD_ARG;
  if (geom1_ok(d))
    return launch_gemm1((const bf16_t*)dy, (const bf16_t*)wb, (const bf16_t*)add, (const bf16_t*)gate, (bf16_t*)dx, nullptr,
                        d, d->Cout, d->Cin, 1, (hipStream_t)stream);
  if (!geom_ok(d)) return DRAM_ERR_UNSUPPORTED;
  return launch_conv((const bf16_t*)dy, (const bf16_t*)wb, nullptr, (const bf16_t*)add, (const bf16_t*)gate, (bf16_t*)dx,
                     nullptr, d, d->Cout, d->Cin, 1, (hipStream_t)stream);
}

extern "C" size_t dram_conv3d_bwd_weight_bf16_workspace(const DramConvDesc* d) {
  if (geom1_ok(d)) {
    if (wgrad1_big(d)) {
      W1bGeom wb{};
      plan_wgrad1b(d, wb);
      return (size_t)wb.nsplit * wb.npairs * 4096 * sizeof(float);
    }
    W1Geom w1{};
    plan_wgrad1(d, w1);
    return (size_t)w1.nsplit * w1.npairs * 4096 * sizeof(float);
  }
  if (!geom_ok(d)) return 0;
  if (use_zwalk(d)) {
    ZGeom z{};
    plan_zwalk(d, z);
    return (size_t)z.nsplit * z.npairs * 27 * 4096 * sizeof(float);
  }
  WGeom g{};
  plan_wgrad(d, g);
  return (size_t)g.nsplit * g.npairs * 27 * 2048 * sizeof(float);
}

extern "C" int dram_conv3d_bwd_weight_bf16(const void* x, const void* dy, float* dw, const DramConvDesc* d,
                                           void* workspace, size_t workspace_bytes, dram_stream_t stream) {
  if (!x || !dy || !dw) return DRAM_ERR_BAD_ARG;
  if (geom1_ok(d) && wgrad1_big(d)) {
    W1bGeom wb{};
    plan_wgrad1b(d, wb);
    const size_t needb = (size_t)wb.nsplit * wb.npairs * 4096 * sizeof(float);
    if (!workspace || workspace_bytes < needb) return DRAM_ERR_WORKSPACE;
    hipStream_t s1 = (hipStream_t)stream;
    {
      DramProf prof(DRAM_FAM_WGRAD_BF16, 6, 2.0 * (double)wb.ntile * 128.0 * d->Cin * d->Cout,
                    2.0 * (double)wb.M * (d->Cin + d->Cout) + 4.0 * d->Cin * d->Cout, s1, 2.0 * (double)wb.M * d->Cin * d->Cout);
      hipLaunchKernelGGL(wgrad1b_bf16_kernel, dim3(wb.nblk), dim3(256), 0, s1, (const bf16_t*)x, (const bf16_t*)dy,
                         (float*)workspace, wb);
      DRAM_LAUNCH_CHECK();
    }
    const long n1 = (long)d->Cout * d->Cin;
    DramProf prof(DRAM_FAM_WGRAD_BF16, 5, 0.0, 4.0 * (double)n1 * (wb.nsplit + 1), s1);
    launch_reduce((const float*)workspace, dw, d->Cout, d->Cin, wb.ci_blocks, wb.npairs, wb.nsplit, 64, 1, s1);
    DRAM_LAUNCH_CHECK();
    return DRAM_OK;
  }
  if (geom1_ok(d)) {
    W1Geom w1{};
    plan_wgrad1(d, w1);
    const size_t need1 = (size_t)w1.nsplit * w1.npairs * 4096 * sizeof(float);
    if (!workspace || workspace_bytes < need1) return DRAM_ERR_WORKSPACE;
    hipStream_t s1 = (hipStream_t)stream;
    {
      DramProf prof(DRAM_FAM_WGRAD_BF16, 4, 2.0 * (double)w1.ntile * 256.0 * d->Cin * d->Cout,
                    2.0 * (double)w1.M * (d->Cin + d->Cout) + 4.0 * d->Cin * d->Cout, s1, 2.0 * (double)w1.M * d->Cin * d->Cout);
      hipLaunchKernelGGL(wgrad1_bf16_kernel, dim3(w1.nblk), dim3(256), 0, s1, (const bf16_t*)x, (const bf16_t*)dy,
                         (float*)workspace, w1);
      DRAM_LAUNCH_CHECK();
    }
    const long n1 = (long)d->Cout * d->Cin;
    DramProf prof(DRAM_FAM_WGRAD_BF16, 5, 0.0, 4.0 * (double)n1 * (w1.nsplit + 1), s1);
    launch_reduce((const float*)workspace, dw, d->Cout, d->Cin, w1.ci_blocks, w1.npairs, w1.nsplit, 64, 1, s1);
    DRAM_LAUNCH_CHECK();
    return DRAM_OK;
  }
  if (!geom_ok(d)) return DRAM_ERR_UNSUPPORTED;
  if (use_zwalk(d)) {
    ZGeom z{};
    plan_zwalk(d, z);
    const size_t needz = (size_t)z.nsplit * z.npairs * 27 * 4096 * sizeof(float);
    if (!workspace || workspace_bytes < needz) return DRAM_ERR_WORKSPACE;
    hipStream_t sz = (hipStream_t)stream;
    const double voxz = (double)d->B * d->D * d->H * d->W;
    {
      DramProf prof(DRAM_FAM_WGRAD_BF16, 2, 2.0 * voxz * 27.0 * d->Cin * (d->Cout <= 32 ? 32.0 : z.co_blocks * 64.0),
                    2.0 * voxz * (d->Cin + d->Cout) + 4.0 * 27.0 * d->Cin * d->Cout, sz, 2.0 * voxz * 27.0 * d->Cin * d->Cout);
      if (d->Cout <= 32)
        hipLaunchKernelGGL(wgrad3z_bf16_kernel<1>, dim3(z.nblk), dim3(512), 0, sz, (const bf16_t*)x, (const bf16_t*)dy,
                           (float*)workspace, z);
      else
        hipLaunchKernelGGL(wgrad3z_bf16_kernel<2>, dim3(z.nblk), dim3(512), 0, sz, (const bf16_t*)x, (const bf16_t*)dy,
                           (float*)workspace, z);
      DRAM_LAUNCH_CHECK();
    }
    const long nz = (long)d->Cout * d->Cin * 27;
    DramProf prof(DRAM_FAM_WGRAD_BF16, 3, 0.0, 4.0 * (double)nz * (z.nsplit + 1), sz);
    launch_reduce((const float*)workspace, dw, d->Cout, d->Cin, z.ci_blocks, z.npairs, z.nsplit, 64, 27, sz);
    DRAM_LAUNCH_CHECK();
    return DRAM_OK;
  }
  WGeom g{};
  plan_wgrad(d, g);
  const size_t need = (size_t)g.nsplit * g.npairs * 27 * 2048 * sizeof(float);
  if (!workspace || workspace_bytes < need) return DRAM_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const double vox = (double)d->B * d->D * d->H * d->W;
  {
    DramProf prof(DRAM_FAM_WGRAD_BF16, 0, 2.0 * (double)g.ntile * 256.0 * 27.0 * g.ci_blocks * 32.0 * g.co_blocks * 64.0,
                  2.0 * vox * (d->Cin + d->Cout) + 4.0 * 27.0 * d->Cin * d->Cout, s, 2.0 * vox * 27.0 * d->Cin * d->Cout);
    static const bool old_form = tune_env("DRAM_BF16_WGRAD_TILE") && !strcmp(tune_env("DRAM_BF16_WGRAD_TILE"), "old");   // A/B
    if (old_form)
      hipLaunchKernelGGL(wgrad3_bf16_kernel, dim3(g.nblk), dim3(512), 0, s, (const bf16_t*)x, (const bf16_t*)dy,
                         (float*)workspace, g);
    else
      hipLaunchKernelGGL(wgrad3b_bf16_kernel, dim3(g.nblk), dim3(512), 0, s, (const bf16_t*)x, (const bf16_t*)dy,
                         (float*)workspace, g);
    DRAM_LAUNCH_CHECK();
  }
  const long n = (long)d->Cout * d->Cin * 27;
  DramProf prof(DRAM_FAM_WGRAD_BF16, 1, 0.0, 4.0 * (double)n * (g.nsplit + 1), s);
  launch_reduce((const float*)workspace, dw, d->Cout, d->Cin, g.ci_blocks, g.npairs, g.nsplit, 32, 27, s);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
