// bn.hip -- BatchNorm3d (+ReLU, + residual) forward/backward, column reductions, add.
// Replaces nn.BatchNorm3d / nn.ReLU / `out += residual` / downsample_basic_block at
// reference med3d.py:121-124, :133-142, :153-182, :103-112, :203-204, :227-228 and their
// autograd backward.  All HBM-bound: float4 (4 channels) per lane, NDHWC rows.
//
// Training statistics arrive as per-tile partial sums from the conv epilogue
// (conv_igemm.hip / stem.hip); dram_reduce_partials folds them in double precision, the
// caller may all-reduce the [2][C] doubles across ranks (SyncBatchNorm), then
// dram_bn_finalize produces mean / invstd / fused scale+shift and the running-stat update.
#include <stdlib.h>
#include <atomic>
#include "common.h"

namespace {

// sums[col] = sum_p partial[p][col] in double.  Stage 1: grid (cols/64, S) -- block (x, s)
// folds parts p = s, s+S, ... (4 interleaved lanes per column through LDS) into tmp[s][col];
// stage 2 (same kernel, double input, S = 1) folds the S rows.  Fixed order -> deterministic.
template <typename T>
__global__ void reduce_partials_kernel(const T* __restrict__ partial, double* __restrict__ out, int nparts,
                                       int RC, double tail, int has_tail) {
  __shared__ double sm[256];
  const int col = blockIdx.x * 64 + (threadIdx.x & 63);
  const int lane4 = threadIdx.x >> 6;  // 0..3
  const int S = gridDim.y, sidx = blockIdx.y;
  double s = 0.0;
  if (col < RC)
    for (int p = sidx + S * lane4; p < nparts; p += 4 * S) s += (double)partial[(long)p * RC + col];
  sm[threadIdx.x] = s;
  __syncthreads();
  if (lane4 == 0 && col < RC)
    out[(long)sidx * RC + col] = sm[threadIdx.x] + sm[threadIdx.x + 64] + sm[threadIdx.x + 128] + sm[threadIdx.x + 192];
  // SyncBN: the rank's element count rides behind the sums, so ONE all-reduce yields global sums and count
  if (has_tail && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) out[RC] = tail;
}

__global__ void bn_finalize_kernel(const double* __restrict__ sums, double count_host,
                                   const double* __restrict__ count_dev, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* __restrict__ rmean,
                                   float* __restrict__ rvar, float momentum, float eps, int update,
                                   float* __restrict__ mean, float* __restrict__ invstd, float* __restrict__ scale,
                                   float* __restrict__ shift, int C) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double m, v;
  const double count = count_dev ? *count_dev : count_host;
  if (sums) {
    m = sums[c] / count;
    v = sums[C + c] / count - m * m;
    if (v < 0.0) v = 0.0;
    if (update) {
      const double unb = count > 1.0 ? v * (count / (count - 1.0)) : v;
      rmean[c] = (float)((1.0 - (double)momentum) * (double)rmean[c] + (double)momentum * m);
      rvar[c] = (float)((1.0 - (double)momentum) * (double)rvar[c] + (double)momentum * unb);
    }
  } else {
    m = (double)rmean[c];
    v = (double)rvar[c];
  }
  const double is = 1.0 / sqrt(v + (double)eps);
  const float sc = (float)((double)gamma[c] * is);
  mean[c] = (float)m;
  invstd[c] = (float)is;
  scale[c] = sc;
  shift[c] = (float)((double)beta[c] - m * (double)gamma[c] * is);
}

// One-launch fold (+ BatchNorm finalize): the stage-1 blocks of reduce_partials_kernel, and the LAST block to finish
// (a ticket counter) folds the S stage rows in fixed order -- deterministic whichever block that is -- writes the
// double sums, optionally a float copy (the BatchNorm parameter gradients), and optionally runs bn_finalize on them.
// A ResNet-50 step issued 114 + 76 + 54 launches of 4-6 us for what is 108 launches now.
struct FoldFinalize {               // gamma == NULL: no finalize
  double count;
  const float* gamma; const float* beta; float* rmean; float* rvar;
  float momentum, eps; int update;
  float* mean; float* invstd; float* scale; float* shift;
};
// S > 1: the tickets are PER-CALL memory (FOLD_XMAX words behind the S stage rows of `scratch`, zeroed by a memset
// in front of the launch): launches in flight together -- two streams, a graph replay beside an eager step, two host
// threads -- cannot share a counter, whatever their order.
constexpr int FOLD_XMAX = 512;

// grid (column groups, S stage rows), 1 024 threads = 16 part lanes x 64 columns.  A block owns 64 columns: plain mode 64 consecutive ones; finalize mode the
// sum AND the sum of squares of 32 channels (columns c and C + c), so that the block which completes a column group
// can finalize its channels by itself.  S == 1: every block is complete on its own (no ticket).  S > 1: one ticket
// per column group, the last of its S blocks folds the S stage rows (independent agent-scope loads, fixed order).
__global__ __launch_bounds__(1024) void fold_partials_kernel(const float* __restrict__ partial, double* scratch,
                                                             double* __restrict__ sums, float* __restrict__ sums_f32,
                                                             float* __restrict__ f32_row1, const int nparts, const int RC,
                                                             const double tail,
                                                             const int has_tail, unsigned* __restrict__ ticket,
                                                             const FoldFinalize ff) {
  __shared__ double sm[1024];
  __shared__ double tot[64];
  __shared__ unsigned s_ticket;
  const int t64 = threadIdx.x & 63, lane4 = threadIdx.x >> 6;      // 16 part lanes x 64 columns
  const int C = RC >> 1;
  const int col = ff.gamma ? (t64 >> 5) * C + blockIdx.x * 32 + (t64 & 31) : blockIdx.x * 64 + t64;
  const bool live = ff.gamma ? (blockIdx.x * 32 + (t64 & 31) < C) : (col < RC);
  const int S = gridDim.y, sidx = blockIdx.y;
  double s = 0.0;
  if (live) {
    // eight independent loads in flight per thread, added in the same fixed order as a plain loop (a loop with
    // the add in it waits out one memory latency per row: up to 64 of them in a row at 1 024 parts)
    const int step = 16 * S;
    int p = sidx + S * lane4;
    for (; p + 7 * step < nparts; p += 8 * step) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = partial[(long)(p + j * step) * RC + col];
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (double)v[j];
    }
    for (; p < nparts; p += step) s += (double)partial[(long)p * RC + col];
  }
  sm[threadIdx.x] = s;
  __syncthreads();
  double total = 0.0;
  if (lane4 == 0) {
#pragma unroll
    for (int j = 0; j < 16; ++j) total += sm[t64 + 64 * j];        // fixed order
  }
  if (S > 1) {
    if (lane4 == 0 && live) scratch[(long)sidx * RC + col] = total;
    __threadfence();                                 // this block's stage row is visible device-wide ...
    __syncthreads();
    if (threadIdx.x == 0) s_ticket = atomicAdd(&ticket[blockIdx.x], 1u);   // ... before its ticket
    __syncthreads();
    if (s_ticket != (unsigned)S - 1) return;
    __threadfence();
    if (lane4 == 0 && live) {
      total = 0.0;
      int k = 0;
      for (; k + 8 <= S; k += 8) {                   // written by other CUs: agent-scope loads, eight in flight
        double v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j)
          v[j] = __hip_atomic_load(scratch + (long)(k + j) * RC + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < 8; ++j) total += v[j];
      }
      for (; k < S; ++k)
        total += __hip_atomic_load(scratch + (long)k * RC + col, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  if (lane4 == 0 && live) {
    sums[col] = total;
    // float copy (BatchNorm parameter gradients): row 1 may live in its own allocation, so that both rows are
    // whole tensors autograd can take over without a copy (a row VIEW of one [2][C] tensor is cloned by AccumulateGrad)
    if (sums_f32) {
      if (f32_row1 && col >= C) f32_row1[col - C] = (float)total;
      else sums_f32[col] = (float)total;
    }
  }
  if (has_tail && blockIdx.x == 0 && threadIdx.x == 0) sums[RC] = tail;
  if (!ff.gamma) return;
  if (lane4 == 0) tot[t64] = total;
  __syncthreads();
  const int c = blockIdx.x * 32 + threadIdx.x;
  if (threadIdx.x < 32 && c < C) {
    const double m = tot[threadIdx.x] / ff.count;
    double v = tot[32 + threadIdx.x] / ff.count - m * m;
    if (v < 0.0) v = 0.0;
    if (ff.update) {
      const double unb = ff.count > 1.0 ? v * (ff.count / (ff.count - 1.0)) : v;
      ff.rmean[c] = (float)((1.0 - (double)ff.momentum) * (double)ff.rmean[c] + (double)ff.momentum * m);
      ff.rvar[c] = (float)((1.0 - (double)ff.momentum) * (double)ff.rvar[c] + (double)ff.momentum * unb);
    }
    const double is = 1.0 / sqrt(v + (double)ff.eps);
    ff.mean[c] = (float)m;
    ff.invstd[c] = (float)is;
    ff.scale[c] = (float)((double)ff.gamma[c] * is);
    ff.shift[c] = (float)((double)ff.beta[c] - m * (double)ff.gamma[c] * is);
  }
}

// z = act(y*scale + shift + residual)
template <int RES, typename T>  // 0 none, 1 same-shape identity (shortcut A: bn_apply_shortcut_a_kernel)
__global__ void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                const float* __restrict__ shift, const T* __restrict__ res, int Cr, int rs,
                                T* __restrict__ z, int D, int H, int W, int C, long total4, int relu) {
  const int Q = C >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * q);
    const float4 sh = *reinterpret_cast<const float4*>(shift + 4 * q);
    const float4 v = ld4<T>(y, 4 * i);
    float4 o;
    // explicit fma: the backward pass re-derives the ReLU mask from y with the same expression (bitwise)
    o.x = __builtin_fmaf(v.x, sc.x, sh.x); o.y = __builtin_fmaf(v.y, sc.y, sh.y);
    o.z = __builtin_fmaf(v.z, sc.z, sh.z); o.w = __builtin_fmaf(v.w, sc.w, sh.w);
    if (RES == 1) {
      const float4 rr = ld4<T>(res, 4 * i);
      o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    st4<T>(z, 4 * i, o);
  }
}

// The same pass for channel counts whose quad count divides the block size (every network width but 576): a
// thread's channel quad is then fixed across its grid-stride loop (no 64-bit modulo, scale / shift in registers),
// and U independent 16-byte (bf16: 8-byte) loads per tensor are issued before the first use -- the plain loop has
// one load per thread in flight, ~8 MB chip-wide, less than the HBM latency-bandwidth product.
template <int RES, typename T, int U>
__global__ __launch_bounds__(256) void bn_apply_fast_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const T* __restrict__ res,
                                                            T* __restrict__ z, int Q, long total4, int relu) {
  const int q = threadIdx.x % Q;
  const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * q);
  const float4 sh = *reinterpret_cast<const float4*>(shift + 4 * q);
  const long stride = (long)gridDim.x * 256;
  for (long i0 = blockIdx.x * 256L + threadIdx.x; i0 < total4; i0 += U * stride) {
    float4 v[U], rr[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < total4) {
        v[u] = ld4<T>(y, 4 * i);
        if (RES == 1) rr[u] = ld4<T>(res, 4 * i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < total4) {
        float4 o;
        o.x = __builtin_fmaf(v[u].x, sc.x, sh.x); o.y = __builtin_fmaf(v[u].y, sc.y, sh.y);
        o.z = __builtin_fmaf(v[u].z, sc.z, sh.z); o.w = __builtin_fmaf(v[u].w, sc.w, sh.w);
        if (RES == 1) { o.x += rr[u].x; o.y += rr[u].y; o.z += rr[u].z; o.w += rr[u].w; }
        if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
        st4<T>(z, 4 * i, o);
      }
    }
  }
}

// The same pass as ONE-SHOT blocks (common.h, ew_blocks): block b owns the 1024 channel quads [1024 b, 1024 (b + 1)),
// four per thread 256 apart (the thread's channel quad stays fixed: 256 % Q == 0).  NT: streaming cache policy on the
// tensor loads and stores.  Same expression per element (bit-identical results).
template <int RES, typename T, bool NT>
__global__ __launch_bounds__(256) void bn_apply_shot_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, const T* __restrict__ res,
                                                            T* __restrict__ z, int Q, long total4, int relu) {
  const int q = threadIdx.x % Q;
  const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * q);
  const float4 sh = *reinterpret_cast<const float4*>(shift + 4 * q);
  const long i0 = blockIdx.x * 1024L + threadIdx.x;
  float4 v[4], rr[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = i0 + u * 256;
    if (i < total4) {
      v[u] = NT ? ld4s<T>(y, 4 * i) : ld4<T>(y, 4 * i);
      if (RES == 1) rr[u] = NT ? ld4s<T>(res, 4 * i) : ld4<T>(res, 4 * i);
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const long i = i0 + u * 256;
    if (i < total4) {
      float4 o;
      o.x = __builtin_fmaf(v[u].x, sc.x, sh.x); o.y = __builtin_fmaf(v[u].y, sc.y, sh.y);
      o.z = __builtin_fmaf(v[u].z, sc.z, sh.z); o.w = __builtin_fmaf(v[u].w, sc.w, sh.w);
      if (RES == 1) { o.x += rr[u].x; o.y += rr[u].y; o.z += rr[u].z; o.w += rr[u].w; }
      if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      if (NT) st4s<T>(z, 4 * i, o); else st4<T>(z, 4 * i, o);
    }
  }
}

// shortcut type A needs the residual tensor's own dims -> dedicated kernel
template <typename T>
__global__ void bn_apply_shortcut_a_kernel(const T* __restrict__ y, const float* __restrict__ scale,
                                           const float* __restrict__ shift, const T* __restrict__ res,
                                           int Dr, int Hr, int Wr, int Cr, int rs, T* __restrict__ z, int D,
                                           int H, int W, int C, long total4, int relu) {
  const int Q = C >> 2;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int q = (int)(i % Q);
    const float4 sc = *reinterpret_cast<const float4*>(scale + 4 * q);
    const float4 sh = *reinterpret_cast<const float4*>(shift + 4 * q);
    const float4 v = ld4<T>(y, 4 * i);
    float4 o;
    o.x = v.x * sc.x + sh.x; o.y = v.y * sc.y + sh.y; o.z = v.z * sc.z + sh.z; o.w = v.w * sc.w + sh.w;
    if (4 * q < Cr) {
      long vox = i / Q;
      const int xx = (int)(vox % W); vox /= W;
      const int yy = (int)(vox % H); vox /= H;
      const int zz = (int)(vox % D);
      const long b = vox / D;
      const long ro = ((((b * Dr + (long)zz * rs) * Hr + (long)yy * rs) * Wr + (long)xx * rs) * Cr) + 4 * q;
      const float4 rr = ld4<T>(res, ro);
      o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
    }
    if (relu) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
    st4<T>(z, 4 * i, o);
  }
}

// Column reductions over rows of an [rows][C] tensor.
// MODE 0: partial[p][0][c] = sum a          (R = 1)
// MODE 1: BN backward: g = dz*(z>0); partial[p][0][c] = sum g, partial[p][1][c] = sum g*xhat (R = 2)
// U: rows per batch of loads (1: few registers, occupancy hides the latency -- the large tensors; 8: a grid of at most a
// few workgroups per CU, where the rows in flight per thread are all the memory-level parallelism there is).
template <int MODE, typename T, int U = 1>
__global__ __launch_bounds__(256) void colreduce_kernel(const T* __restrict__ a, const T* __restrict__ zz,
                                                        const T* __restrict__ yy, const float* __restrict__ mean,
                                                        const float* __restrict__ invstd, float* __restrict__ partial,
                                                        long rows, int C, int rpb, int relu,
                                                        const float* __restrict__ scale = nullptr,
                                                        const float* __restrict__ shift = nullptr) {
  constexpr int R = MODE == 0 ? 1 : 2;
  __shared__ float4 sm[R][256];
  const int Q = C >> 2;
  const int lanes = Q < 256 ? Q : 256;
  const int rg = 256 / lanes;
  const int tid = threadIdx.x;
  const int ql = tid % lanes, grp = tid / lanes;
  const long r0 = (long)blockIdx.x * rpb;
  const long r1 = (r0 + rpb < rows) ? r0 + rpb : rows;
  // (C > 1 024: the 256-lane channel chunks are blocks of their own -- gridDim.y -- instead of a loop: twice / four
  // times the workgroups on the 16 x 32 x 32 stages of ResNet-50, whose row count gives one workgroup per CU)
  for (int qb = blockIdx.y * lanes; qb < Q; qb += gridDim.y * lanes) {
    const int q = qb + ql;
    float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
    if (grp < rg && q < Q) {
      float4 mu, is, sc, sh;
      if (MODE == 1) {
        mu = *reinterpret_cast<const float4*>(mean + 4 * q);
        is = *reinterpret_cast<const float4*>(invstd + 4 * q);
        if (relu && !zz) {
          sc = *reinterpret_cast<const float4*>(scale + 4 * q);
          sh = *reinterpret_cast<const float4*>(shift + 4 * q);
        }
      }
      // Rows in batches of U with every load of the batch issued before the first add (same order of additions as a
      // plain loop: bit-identical sums).  The plain loop compiled to load - wait - add per row: one row in flight per
      // thread, and on the wide 16 x 32 x 32 stages of ResNet-50 (512-2 048 channels, 64 rows per block, ONE workgroup
      // per CU) that is 64 memory latencies in a row -- 92 us for 134 MB (2.2 TB/s in bf16; 1.2 TB/s on the 4-MB tensors).
      auto accumulate = [&](float4 g, const float4 yv, const float4 zraw) __attribute__((always_inline)) {
        if (MODE == 1) {
          if (relu) {
            // mask of the forward ReLU: from the saved output z, or (no residual) re-derived from y with the
            // forward's own expression -- one tensor read less
            float4 zv = zraw;
            if (!zz) zv = make_float4(__builtin_fmaf(yv.x, sc.x, sh.x), __builtin_fmaf(yv.y, sc.y, sh.y),
                                      __builtin_fmaf(yv.z, sc.z, sh.z), __builtin_fmaf(yv.w, sc.w, sh.w));
            g.x = zv.x > 0.f ? g.x : 0.f; g.y = zv.y > 0.f ? g.y : 0.f;
            g.z = zv.z > 0.f ? g.z : 0.f; g.w = zv.w > 0.f ? g.w : 0.f;
          }
          s1.x += g.x * ((yv.x - mu.x) * is.x); s1.y += g.y * ((yv.y - mu.y) * is.y);
          s1.z += g.z * ((yv.z - mu.z) * is.z); s1.w += g.w * ((yv.w - mu.w) * is.w);
        }
        s0.x += g.x; s0.y += g.y; s0.z += g.z; s0.w += g.w;
      };
      long r = r0 + grp;
      for (; r + (long)(U - 1) * rg < r1; r += (long)U * rg) {
        float4 gv[U], yv[U], zv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
          const long o = (r + (long)u * rg) * C + 4 * q;
          gv[u] = ld4<T>(a, o);
          if (MODE == 1) {
            yv[u] = ld4<T>(yy, o);
            if (relu && zz) zv[u] = ld4<T>(zz, o);
          }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) accumulate(gv[u], yv[u], zv[u]);
      }
      for (; r < r1; r += rg) {
        const long o = r * C + 4 * q;
        const float4 g = ld4<T>(a, o);
        float4 yv = g, zv = g;
        if (MODE == 1) {
          yv = ld4<T>(yy, o);
          if (relu && zz) zv = ld4<T>(zz, o);
        }
        accumulate(g, yv, zv);
      }
    }
    __syncthreads();
    sm[0][tid] = s0;
    if (R == 2) sm[R - 1][tid] = s1;
    __syncthreads();
    if (grp == 0 && q < Q) {
#pragma unroll
      for (int rr = 0; rr < R; ++rr) {
        float4 t = sm[rr][ql];
        for (int k = 1; k < rg; ++k) {
          const float4 u = sm[rr][k * lanes + ql];
          t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
        }
        *reinterpret_cast<float4*>(partial + ((long)blockIdx.x * R + rr) * C + 4 * q) = t;
      }
    }
  }
}

template <typename T>
__global__ void bn_bwd_apply_kernel(const T* __restrict__ dz, const T* __restrict__ z,
                                    const T* __restrict__ y, const float* __restrict__ mean,
                                    const float* __restrict__ invstd, const float* __restrict__ gamma,
                                    const double* __restrict__ sums, double inv_count_host,
                                    const double* __restrict__ count_dev, T* __restrict__ dy, int C,
                                    long total4, int relu, const float* __restrict__ scale,
                                    const float* __restrict__ shift, float* __restrict__ colpart) {
  const int Q = C >> 2;
  const double inv_count = count_dev ? 1.0 / *count_dev : inv_count_host;
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);   // colpart: this thread's channel quad is fixed (256 % Q == 0)
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(i % Q);
    float4 g = ld4<T>(dz, 4 * i);
    const float4 yv = ld4<T>(y, 4 * i);
    if (relu) {
      float4 zv;
      if (z) zv = ld4<T>(z, 4 * i);
      else {   // no residual: the forward mask re-derived from y (same fma as bn_apply_kernel)
        const float4 sc = *reinterpret_cast<const float4*>(scale + c);
        const float4 sh = *reinterpret_cast<const float4*>(shift + c);
        zv = make_float4(__builtin_fmaf(yv.x, sc.x, sh.x), __builtin_fmaf(yv.y, sc.y, sh.y),
                         __builtin_fmaf(yv.z, sc.z, sh.z), __builtin_fmaf(yv.w, sc.w, sh.w));
      }
      g.x = zv.x > 0.f ? g.x : 0.f; g.y = zv.y > 0.f ? g.y : 0.f;
      g.z = zv.z > 0.f ? g.z : 0.f; g.w = zv.w > 0.f ? g.w : 0.f;
    }
    const float4 mu = *reinterpret_cast<const float4*>(mean + c);
    const float4 is = *reinterpret_cast<const float4*>(invstd + c);
    const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
    float4 o;
#define BN_BWD_1(f, k)                                               \
  {                                                                   \
    const float mg = (float)(sums[c + k] * inv_count);                \
    const float mgx = (float)(sums[C + c + k] * inv_count);           \
    const float xh = (yv.f - mu.f) * is.f;                            \
    o.f = ga.f * is.f * (g.f - mg - xh * mgx);                        \
  }
    BN_BWD_1(x, 0) BN_BWD_1(y, 1) BN_BWD_1(z, 2) BN_BWD_1(w, 3)
#undef BN_BWD_1
    st4<T>(dy, 4 * i, o);
    cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
  }
  if (colpart) {   // per-block column sums of dy: the gradient of the bias of the convolution in front
    __shared__ float4 sm[256];
    sm[threadIdx.x] = cs;
    __syncthreads();
    if ((int)threadIdx.x < Q) {
      float4 t = sm[threadIdx.x];
      for (int k = threadIdx.x + Q; k < 256; k += Q) {
        const float4 u = sm[k];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      *reinterpret_cast<float4*>(colpart + (long)blockIdx.x * C + 4 * threadIdx.x) = t;
    }
  }
}

// bn_bwd_apply for channel counts whose quad count divides the block size: the thread's channel quad is fixed, so the
// per-channel constants (mean, invstd, gamma, the two batch means -- double multiplies in the generic kernel, per
// element) live in registers, there is no 64-bit modulo per element, and two elements are in flight per thread.
// Same expression per element as bn_bwd_apply_kernel (bit-identical results).
// SHOT 0: capped grid + grid-stride loop; SHOT 1 / 2: one-shot blocks of 256 U K quads (common.h, ew_blocks), 2 =
// streaming loads and stores.  Per-block column sums of dy (the bias gradient of a convolution in front) need few
// enough blocks for the partial rows: those launches sweep K > 1 consecutive 256 U-quad pieces per block (4,096 quads:
// 5.3 TB/s in tools/stream_probe.hip's three-stream form against 5.7 for K = 1 and 4.1-4.6 for the capped grid).  U elements in flight per thread: 2 for fp32, 4 for bf16 (half the bytes per element against the same
// per-thread set-up of the channel constants).
template <typename T, int SHOT, int U, int K>
__global__ __launch_bounds__(256) void bn_bwd_apply_fast_kernel(const T* __restrict__ dz, const T* __restrict__ z,
                                                                const T* __restrict__ y, const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const double* __restrict__ sums, double inv_count_host,
                                                                const double* __restrict__ count_dev,
                                                                T* __restrict__ dy, int C, long total4, int relu,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                float* __restrict__ colpart) {
  const int Q = C >> 2;
  const int c = 4 * (threadIdx.x % Q);
  const double inv_count = count_dev ? 1.0 / *count_dev : inv_count_host;
  const float4 mu = *reinterpret_cast<const float4*>(mean + c);
  const float4 is = *reinterpret_cast<const float4*>(invstd + c);
  const float4 ga = *reinterpret_cast<const float4*>(gamma + c);
  float4 sc = make_float4(0.f, 0.f, 0.f, 0.f), sh = sc;
  if (relu && !z) {
    sc = *reinterpret_cast<const float4*>(scale + c);
    sh = *reinterpret_cast<const float4*>(shift + c);
  }
  const float4 mg = make_float4((float)(sums[c] * inv_count), (float)(sums[c + 1] * inv_count),
                                (float)(sums[c + 2] * inv_count), (float)(sums[c + 3] * inv_count));
  const float4 mgx = make_float4((float)(sums[C + c] * inv_count), (float)(sums[C + c + 1] * inv_count),
                                 (float)(sums[C + c + 2] * inv_count), (float)(sums[C + c + 3] * inv_count));
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
  const long stride = SHOT ? 256L : (long)gridDim.x * 256;
  int kk = 0;
  for (long i0 = blockIdx.x * (SHOT ? 256L * U * K : 256L) + threadIdx.x; i0 < total4; i0 += U * stride) {
    float4 g[U], yv[U], zv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < total4) {
        g[u] = SHOT == 2 ? ld4s<T>(dz, 4 * i) : ld4<T>(dz, 4 * i);
        yv[u] = SHOT == 2 ? ld4s<T>(y, 4 * i) : ld4<T>(y, 4 * i);
        if (relu && z) zv[u] = SHOT == 2 ? ld4s<T>(z, 4 * i) : ld4<T>(z, 4 * i);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long i = i0 + u * stride;
      if (i < total4) {
        float4 gg = g[u];
        if (relu) {
          float4 m = zv[u];
          if (!z) m = make_float4(__builtin_fmaf(yv[u].x, sc.x, sh.x), __builtin_fmaf(yv[u].y, sc.y, sh.y),
                                  __builtin_fmaf(yv[u].z, sc.z, sh.z), __builtin_fmaf(yv[u].w, sc.w, sh.w));
          gg.x = m.x > 0.f ? gg.x : 0.f; gg.y = m.y > 0.f ? gg.y : 0.f;
          gg.z = m.z > 0.f ? gg.z : 0.f; gg.w = m.w > 0.f ? gg.w : 0.f;
        }
        float4 o;
        o.x = ga.x * is.x * (gg.x - mg.x - ((yv[u].x - mu.x) * is.x) * mgx.x);
        o.y = ga.y * is.y * (gg.y - mg.y - ((yv[u].y - mu.y) * is.y) * mgx.y);
        o.z = ga.z * is.z * (gg.z - mg.z - ((yv[u].z - mu.z) * is.z) * mgx.z);
        o.w = ga.w * is.w * (gg.w - mg.w - ((yv[u].w - mu.w) * is.w) * mgx.w);
        if (SHOT == 2) st4s<T>(dy, 4 * i, o); else st4<T>(dy, 4 * i, o);
        cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
      }
    }
    if (SHOT && ++kk == K) break;
  }
  if ((SHOT == 0 || K > 1) && colpart) {
    __shared__ float4 sm[256];
    sm[threadIdx.x] = cs;
    __syncthreads();
    if ((int)threadIdx.x < Q) {
      float4 t = sm[threadIdx.x];
      for (int k = threadIdx.x + Q; k < 256; k += Q) {
        const float4 u = sm[k];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
      *reinterpret_cast<float4*>(colpart + (long)blockIdx.x * C + 4 * threadIdx.x) = t;
    }
  }
}

__global__ void add_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o,
                           long n4, const float* __restrict__ as, const float* __restrict__ bs,
                           float* __restrict__ os, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 u = a[i], v = b[i];
    o[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
  // tail (n not a multiple of 4)
  const long t0 = n4 * 4;
  for (long i = t0 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    os[i] = as[i] + bs[i];
}

inline int ew_grid(long total4) {
  long b = (total4 + 255) / 256;
  return (int)(b > 4096 ? 4096 : (b < 1 ? 1 : b));
}

inline int rows_per_block(long long rows) {
  long long rpb = (rows + 1023) / 1024;
  if (rpb < 64) rpb = 64;
  return (int)rpb;
}

}  // namespace

extern "C" int dram_reduce_partials_stages(int nparts) {
  // number of stage-1 rows (scratch doubles needed = stages * R * C); 1 => single pass
  if (nparts <= 64) return 1;
  int s = (nparts + 31) / 32;
  return s > 64 ? 64 : s;
}

extern "C" int dram_reduce_partials(const float* partial, double* sums, double* scratch, int nparts, int R, int C,
                                    double tail, int has_tail, dram_stream_t stream) {
  if (!partial || !sums || nparts < 1 || R < 1 || C < 1) return DRAM_ERR_BAD_ARG;
  const int RC = R * C;
  const int S = dram_reduce_partials_stages(nparts);
  hipStream_t st = (hipStream_t)stream;
  DramProf prof(DRAM_FAM_BN, 0, 0.0, 4.0 * (double)nparts * RC + 8.0 * RC, st);
  if (S == 1) {
    hipLaunchKernelGGL((reduce_partials_kernel<float>), dim3((RC + 63) / 64, 1), dim3(256), 0, st, partial, sums,
                       nparts, RC, tail, has_tail);
  } else {
    if (!scratch) return DRAM_ERR_WORKSPACE;
    hipLaunchKernelGGL((reduce_partials_kernel<float>), dim3((RC + 63) / 64, S), dim3(256), 0, st, partial, scratch,
                       nparts, RC, 0.0, 0);
    DRAM_LAUNCH_CHECK();
    hipLaunchKernelGGL((reduce_partials_kernel<double>), dim3((RC + 63) / 64, 1), dim3(256), 0, st,
                       (const double*)scratch, sums, S, RC, tail, has_tail);
  }
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_fold_partials_stages(int nparts) {
  // one block of 16 part lanes folds up to 1 024 partial rows by itself (every layer below the 64 x 128 x 128 grid:
  // no ticket, no second pass); above that, stage rows of 512 parts each
  if (nparts <= 1024) return 1;
  const int s = (nparts + 511) / 512;
  return s > 64 ? 64 : s;
}

static int fold_launch(const float* partial, double* sums, double* scratch, float* sums_f32, float* f32_row1, int nparts,
                       int RC, double tail, int has_tail, const FoldFinalize& ff, hipStream_t st) {
  const int S = dram_fold_partials_stages(nparts);
  const int gx = ff.gamma ? (RC / 2 + 31) / 32 : (RC + 63) / 64;
  if (S > 1 && gx > FOLD_XMAX) return DRAM_ERR_UNSUPPORTED;        // (no such layer: S > 1 goes with few channels)
  unsigned* ticket = nullptr;
  if (S > 1) {                                       // (a memset NODE under capture: every replay starts from zero)
    ticket = reinterpret_cast<unsigned*>(scratch + (long)S * RC);
    const hipError_t e = hipMemsetAsync(ticket, 0, sizeof(unsigned) * gx, st);
    if (e != hipSuccess) return (int)e;
  }
  DramProf prof(DRAM_FAM_BN, 0, 0.0, 4.0 * (double)nparts * RC + 8.0 * RC, st);
  hipLaunchKernelGGL(fold_partials_kernel, dim3(gx, S), dim3(1024), 0, st, partial, scratch, sums, sums_f32, f32_row1,
                     nparts, RC, tail, has_tail, ticket, ff);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_fold_partials(const float* partial, double* sums, double* scratch, float* sums_f32, float* f32_row1,
                                  int nparts, int R, int C, double tail, int has_tail, dram_stream_t stream) {
  if (!partial || !sums || !scratch || nparts < 1 || R < 1 || C < 1) return DRAM_ERR_BAD_ARG;
  if (f32_row1 && (R != 2 || !sums_f32)) return DRAM_ERR_BAD_ARG;
  FoldFinalize ff{};
  return fold_launch(partial, sums, scratch, sums_f32, f32_row1, nparts, R * C, tail, has_tail, ff, (hipStream_t)stream);
}

extern "C" int dram_bn_fold_finalize(const float* partial, double* sums, double* scratch, int nparts, int C, double count,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     float momentum, float eps, int update_running, float* mean, float* invstd,
                                     float* scale, float* shift, dram_stream_t stream) {
  if (!partial || !sums || !scratch || nparts < 1 || C < 1 || count <= 0.0) return DRAM_ERR_BAD_ARG;
  if (!gamma || !beta || !mean || !invstd || !scale || !shift) return DRAM_ERR_BAD_ARG;
  if (update_running && (!running_mean || !running_var)) return DRAM_ERR_BAD_ARG;
  FoldFinalize ff{count, gamma, beta, running_mean, running_var, momentum, eps, update_running, mean, invstd, scale, shift};
  return fold_launch(partial, sums, scratch, nullptr, nullptr, nparts, 2 * C, 0.0, 0, ff, (hipStream_t)stream);
}

extern "C" int dram_bn_finalize(const double* sums, double count, const double* count_dev, const float* gamma,
                                const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps,
                                int update_running, float* mean, float* invstd, float* scale, float* shift, int C,
                                dram_stream_t stream) {
  if (!gamma || !beta || !mean || !invstd || !scale || !shift || C < 1) return DRAM_ERR_BAD_ARG;
  if (!sums && (!running_mean || !running_var)) return DRAM_ERR_BAD_ARG;
  if (sums && !count_dev && count <= 0.0) return DRAM_ERR_BAD_ARG;
  if (update_running && (!running_mean || !running_var)) return DRAM_ERR_BAD_ARG;
  DramProf prof(DRAM_FAM_BN, 1, 0.0, 4.0 * 10.0 * C, (hipStream_t)stream);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, count,
                     count_dev, gamma, beta, running_mean, running_var, momentum, eps, update_running, mean, invstd, scale,
                     shift, C);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

template <typename T>
static int bn_apply_impl(const T* y, const float* scale, const float* shift, const T* residual, int Dr, int Hr, int Wr,
                         int Cr, int rs, T* z, int B, int D, int H, int W, int C, int relu, dram_stream_t stream) {
  if (!y || !scale || !shift || !z || B < 1 || D < 1 || H < 1 || W < 1 || C < 4 || (C & 3)) return DRAM_ERR_BAD_ARG;
  const long total4 = (long)B * D * H * W * (C >> 2);
  hipStream_t s = (hipStream_t)stream;
  const int grid = ew_blocks(total4, 256, 4096);
  // y read, z written, residual read (identity: full size; shortcut A: 1/rs^3 of Cr/C of it)
  const double res_frac = !residual ? 0.0 : ((double)Cr / C) / ((double)rs * rs * rs);
  DramProf prof(DRAM_FAM_BN, 2, 0.0, 4.0 * sizeof(T) * (double)total4 * (2.0 + res_frac), s);
  const int Q = C >> 2;
  const bool identity = residual && rs == 1 && Cr == C && Dr == D && Hr == H && Wr == W;
  static const int ew_u = tune_env("DRAM_EW_U") ? atoi(tune_env("DRAM_EW_U")) : 4;      // A/B switch (tools/ew_bench.py)
  if ((!residual || identity) && 256 % Q == 0 && ew_u > 0 && ew_shape() > 0) {
    const int g1 = ew_blocks(total4, 1024, 0);
#define BN_SHOT_(RES_, NT_) hipLaunchKernelGGL((bn_apply_shot_kernel<RES_, T, NT_>), dim3(g1), dim3(256), 0, s, y, scale, \
                                               shift, residual, z, Q, total4, relu)
    const bool nt = ew_stream(sizeof(T) * 4 * total4);
    if (!residual) { if (nt) BN_SHOT_(0, true); else BN_SHOT_(0, false); }
    else           { if (nt) BN_SHOT_(1, true); else BN_SHOT_(1, false); }
#undef BN_SHOT_
  } else if ((!residual || identity) && 256 % Q == 0 && ew_u > 0) {
    const long per = (total4 + 255) / 256;
    const int g2 = (int)((per + ew_u - 1) / ew_u < 8192 ? ((per + ew_u - 1) / ew_u < 1 ? 1 : (per + ew_u - 1) / ew_u) : 8192);
#define BN_FAST_(RES_, U_) hipLaunchKernelGGL((bn_apply_fast_kernel<RES_, T, U_>), dim3(g2), dim3(256), 0, s, y, scale, shift, \
                                              residual, z, Q, total4, relu)
    if (!residual) { if (ew_u >= 4) BN_FAST_(0, 4); else if (ew_u >= 2) BN_FAST_(0, 2); else BN_FAST_(0, 1); }
    else           { if (ew_u >= 4) BN_FAST_(1, 4); else if (ew_u >= 2) BN_FAST_(1, 2); else BN_FAST_(1, 1); }
#undef BN_FAST_
  } else if (!residual) {
    hipLaunchKernelGGL((bn_apply_kernel<0, T>), dim3(grid), dim3(256), 0, s, y, scale, shift, (const T*)nullptr, 0,
                       1, z, D, H, W, C, total4, relu);
  } else if (identity) {
    hipLaunchKernelGGL((bn_apply_kernel<1, T>), dim3(grid), dim3(256), 0, s, y, scale, shift, residual,
                       Cr, 1, z, D, H, W, C, total4, relu);
  } else {
    // shortcut type A (med3d.py:103-112): strided subsample, channels >= Cr read zero
    if ((Cr & 3) || Cr < 4 || Cr > C || rs < 1) return DRAM_ERR_BAD_ARG;
    if ((D - 1) * rs >= Dr || (H - 1) * rs >= Hr || (W - 1) * rs >= Wr) return DRAM_ERR_BAD_ARG;
    hipLaunchKernelGGL((bn_apply_shortcut_a_kernel<T>), dim3(grid), dim3(256), 0, s, y, scale, shift,
                       residual, Dr, Hr, Wr, Cr, rs, z, D, H, W, C, total4, relu);
  }
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}

extern "C" int dram_bn_apply(const float* y, const float* scale, const float* shift, const float* residual,
                             int Dr, int Hr, int Wr, int Cr, int rs, float* z, int B, int D, int H, int W, int C,
                             int relu, dram_stream_t stream) {
  return bn_apply_impl<float>(y, scale, shift, residual, Dr, Hr, Wr, Cr, rs, z, B, D, H, W, C, relu, stream);
}
extern "C" int dram_bn_apply_bf16(const void* y, const float* scale, const float* shift, const void* residual,
                                  int Dr, int Hr, int Wr, int Cr, int rs, void* z, int B, int D, int H, int W, int C,
                                  int relu, dram_stream_t stream) {
  return bn_apply_impl<bf16_t>((const bf16_t*)y, scale, shift, (const bf16_t*)residual, Dr, Hr, Wr, Cr, rs, (bf16_t*)z,
                               B, D, H, W, C, relu, stream);
}

extern "C" int dram_colsum_nparts(long long rows, int C) {
  if (rows < 1 || C < 4) return DRAM_ERR_BAD_ARG;
  const int rpb = rows_per_block(rows);
  return (int)((rows + rpb - 1) / rpb);
}

template <typename T>
static int bn_bwd_reduce_impl(const T* dz, const T* z, const T* y, const float* mean, const float* invstd,
                              const float* scale, const float* shift, float* partial, long long rows, int C, int relu,
                              dram_stream_t stream) {
  if (!dz || !y || !mean || !invstd || !partial || rows < 1 || C < 4 || (C & 3)) return DRAM_ERR_BAD_ARG;
  if (relu && !z && !(scale && shift)) return DRAM_ERR_BAD_ARG;
  const int rpb = rows_per_block(rows);
  const int nparts = (int)((rows + rpb - 1) / rpb);
  DramProf prof(DRAM_FAM_BN, 3, 0.0, (double)sizeof(T) * (double)rows * C * (relu && z ? 3.0 : 2.0), (hipStream_t)stream);
  const int ychunks = ((C >> 2) + 255) / 256;        // 256-lane channel chunks (1 up to 1 024 channels)
  if (nparts <= 1024)
    hipLaunchKernelGGL((colreduce_kernel<1, T, 8>), dim3(nparts, ychunks), dim3(256), 0, (hipStream_t)stream, dz, z, y, mean,
                       invstd, partial, (long)rows, C, rpb, relu, scale, shift);
  else
    hipLaunchKernelGGL((colreduce_kernel<1, T, 1>), dim3(nparts, ychunks), dim3(256), 0, (hipStream_t)stream, dz, z, y, mean,
                       invstd, partial, (long)rows, C, rpb, relu, scale, shift);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_bn_bwd_reduce(const float* dz, const float* z, const float* y, const float* mean,
                                  const float* invstd, const float* scale, const float* shift, float* partial,
                                  long long rows, int C, int relu, dram_stream_t stream) {
  return bn_bwd_reduce_impl<float>(dz, z, y, mean, invstd, scale, shift, partial, rows, C, relu, stream);
}
extern "C" int dram_bn_bwd_reduce_bf16(const void* dz, const void* z, const void* y, const float* mean,
                                       const float* invstd, const float* scale, const float* shift, float* partial,
                                       long long rows, int C, int relu, dram_stream_t stream) {
  return bn_bwd_reduce_impl<bf16_t>((const bf16_t*)dz, (const bf16_t*)z, (const bf16_t*)y, mean, invstd, scale, shift,
                                    partial, rows, C, relu, stream);
}

template <typename T>
static int colsum_impl(const T* a, float* partial, long long rows, int C, dram_stream_t stream) {
  if (!a || !partial || rows < 1 || C < 4 || (C & 3)) return DRAM_ERR_BAD_ARG;
  const int rpb = rows_per_block(rows);
  const int nparts = (int)((rows + rpb - 1) / rpb);
  DramProf prof(DRAM_FAM_BN, 4, 0.0, (double)sizeof(T) * (double)rows * C, (hipStream_t)stream);
  const int ychunks = ((C >> 2) + 255) / 256;
  if (nparts <= 1024)
    hipLaunchKernelGGL((colreduce_kernel<0, T, 8>), dim3(nparts, ychunks), dim3(256), 0, (hipStream_t)stream, a,
                       (const T*)nullptr, (const T*)nullptr, nullptr, nullptr, partial, (long)rows, C, rpb, 0);
  else
    hipLaunchKernelGGL((colreduce_kernel<0, T, 1>), dim3(nparts, ychunks), dim3(256), 0, (hipStream_t)stream, a,
                       (const T*)nullptr, (const T*)nullptr, nullptr, nullptr, partial, (long)rows, C, rpb, 0);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_colsum(const float* a, float* partial, long long rows, int C, dram_stream_t stream) {
  return colsum_impl<float>(a, partial, rows, C, stream);
}
extern "C" int dram_colsum_bf16(const void* a, float* partial, long long rows, int C, dram_stream_t stream) {
  return colsum_impl<bf16_t>((const bf16_t*)a, partial, rows, C, stream);
}

// rows of colsum_partial written by dram_bn_bwd_apply, or DRAM_ERR_UNSUPPORTED when a thread's channel quad is
// not fixed across its grid-stride loop (C / 4 must divide 256)
extern "C" int dram_bn_bwd_apply_nparts(long long rows, int C) {
  if (rows < 1 || C < 4 || (C & 3)) return DRAM_ERR_BAD_ARG;
  if (256 % (C >> 2) != 0) return DRAM_ERR_UNSUPPORTED;
  if (ew_shape() > 0) return ew_blocks((long)rows * (C >> 2), 4096, 0);    // one row per 4,096-quad block (fast kernel)
  return ew_grid((long)rows * (C >> 2));
}

template <typename T>
static int bn_bwd_apply_impl(const T* dz, const T* z, const T* y, const float* mean, const float* invstd,
                             const float* gamma, const float* scale, const float* shift, const double* sums,
                             double count, const double* count_dev, T* dy, float* colsum_partial, long long rows, int C,
                             int relu, dram_stream_t stream) {
  if (!dz || !y || !mean || !invstd || !gamma || !sums || !dy || rows < 1 || C < 4 || (C & 3) ||
      (!count_dev && count <= 0.0))
    return DRAM_ERR_BAD_ARG;
  if (relu && !z && !(scale && shift)) return DRAM_ERR_BAD_ARG;
  if (colsum_partial && dram_bn_bwd_apply_nparts(rows, C) < 1) return DRAM_ERR_UNSUPPORTED;
  const long total4 = (long)rows * (C >> 2);
  DramProf prof(DRAM_FAM_BN, 5, 0.0, 4.0 * sizeof(T) * (double)total4 * (relu && z ? 4.0 : 3.0), (hipStream_t)stream);
  if (256 % (C >> 2) == 0) {   // (with column sums: same grid as the generic kernel, one partial row per block)
#define BN_BWD_FAST_(SHOT_, U_, K_, GRID_)                                                                                \
  hipLaunchKernelGGL((bn_bwd_apply_fast_kernel<T, SHOT_, U_, K_>), dim3(GRID_), dim3(256), 0, (hipStream_t)stream, dz, z, y, \
                     mean, invstd, gamma, sums, count_dev ? 0.0 : 1.0 / count, count_dev, dy, C, total4, relu, scale,     \
                     shift, colsum_partial)
    constexpr int U = sizeof(T) == 2 ? 4 : 2;
    if (ew_shape() == 0) BN_BWD_FAST_(0, 2, 1, ew_grid(total4));
    else if (colsum_partial) BN_BWD_FAST_(1, U, 16 / U, ew_blocks(total4, 4096, 0));        // = dram_bn_bwd_apply_nparts
    else if (ew_stream(sizeof(T) * 4 * total4)) BN_BWD_FAST_(2, U, 1, ew_blocks(total4, 256 * U, 0));
    else BN_BWD_FAST_(1, U, 1, ew_blocks(total4, 256 * U, 0));
#undef BN_BWD_FAST_
  }
  else
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(ew_grid(total4)), dim3(256), 0, (hipStream_t)stream, dz, z, y, mean,
                       invstd, gamma, sums, count_dev ? 0.0 : 1.0 / count, count_dev, dy, C, total4, relu, scale, shift,
                       colsum_partial);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
extern "C" int dram_bn_bwd_apply(const float* dz, const float* z, const float* y, const float* mean,
                                 const float* invstd, const float* gamma, const float* scale, const float* shift,
                                 const double* sums, double count, const double* count_dev, float* dy,
                                 float* colsum_partial, long long rows, int C, int relu, dram_stream_t stream) {
  return bn_bwd_apply_impl<float>(dz, z, y, mean, invstd, gamma, scale, shift, sums, count, count_dev, dy,
                                  colsum_partial, rows, C, relu, stream);
}
extern "C" int dram_bn_bwd_apply_bf16(const void* dz, const void* z, const void* y, const float* mean,
                                      const float* invstd, const float* gamma, const float* scale, const float* shift,
                                      const double* sums, double count, const double* count_dev, void* dy,
                                      float* colsum_partial, long long rows, int C, int relu, dram_stream_t stream) {
  return bn_bwd_apply_impl<bf16_t>((const bf16_t*)dz, (const bf16_t*)z, (const bf16_t*)y, mean, invstd, gamma, scale,
                                   shift, sums, count, count_dev, (bf16_t*)dy, colsum_partial, rows, C, relu, stream);
}

extern "C" int dram_add(const float* a, const float* b, float* out, long long n, dram_stream_t stream) {
  if (!a || !b || !out || n < 1) return DRAM_ERR_BAD_ARG;
  const long n4 = ((((uintptr_t)a | (uintptr_t)b | (uintptr_t)out) & 15) == 0) ? n / 4 : 0;
  DramProf prof(DRAM_FAM_BN, 6, 0.0, 12.0 * (double)n, (hipStream_t)stream);
  hipLaunchKernelGGL(add_kernel, dim3(ew_blocks(n4 > 0 ? n4 : (n + 3) / 4, 256, 4096)), dim3(256), 0, (hipStream_t)stream,
                     (const float4*)a, (const float4*)b, (float4*)out, n4, a, b, out, (long)n);
  DRAM_LAUNCH_CHECK();
  return DRAM_OK;
}
