/* dram_hip.h -- C ABI of libdram_hip.so (MI355X / gfx950 hand-written HIP kernels).
 *
 * The reference has no FFI of its own: its hot path is torch-op call sites inside
 * med3d.py / models.py / metrics.py.  Each entry point below
 * replaces the cuDNN/ATen kernels behind one family of those call sites; the
 * reference file:line each one stands in for is cited on the declaration.
 *
 * Conventions
 *  - Activations are NDHWC ("channels-last-3D") contiguous float32:
 *      x[b][z][y][x][c], element offset ((((b*D+z)*H+y)*W+x)*C+c).
 *    The reference uses NCDHW; for C == 1 (network input, lung masks, regression
 *    dRAM maps) the two layouts are the same bytes.
 *  - Convolution weights cross the ABI in the reference's own layout
 *    [Cout][Cin][kD][kH][kW] (state_dict contract); packed copies
 *    are produced by dram_pack_conv_weight.
 *  - Every function enqueues on `stream` and never synchronises, allocates or frees.
 *    The caller owns all buffers including workspaces.
 *  - Return value: 0 on success, DRAM_ERR_* (<0) on a rejected argument, or a
 *    positive hipError_t from the launch.  Nothing is thrown across the ABI.
 */
#ifndef DRAM_HIP_H
#define DRAM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dram_stream_t; /* hipStream_t */

#define DRAM_OK 0
#define DRAM_ERR_BAD_ARG (-1)
#define DRAM_ERR_UNSUPPORTED (-2)
#define DRAM_ERR_WORKSPACE (-3)

#define DRAM_ABI_VERSION 7
int dram_version(void);
/* static string: "gfx950" build tag */
const char* dram_build_info(void);
/* sha1 (first 16 hex digits) of the dram_hip.h this library was compiled against; the host binding
 * compares it with the header it reads its signatures from and refuses a mismatching library. */
const char* dram_abi_hash(void);
/* Identity of the hipGraph capture `stream` is recording into (hipStreamGetCaptureInfo): 0 when the stream is not
 * capturing, otherwise a number unique to that capture.  The host side keys everything a captured launch bakes a
 * pointer of (scratch buffers, device work lists) by it, so that no other capture and no eager call can take over or
 * free such memory while the graph lives.  Does not enqueue anything. */
unsigned long long dram_stream_capture_id(dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Kernel timeline (measurement only; off by default, no cost when off).
 * Between dram_profile_start and dram_profile_stop every kernel launch of the library is bracketed by
 * two hipEvents recorded on the stream it is launched on, tagged with its kernel family, the MFMA
 * FLOPs it EXECUTES (products really issued -- a Winograd kernel issues fewer than the direct
 * convolution it replaces) and the ALGORITHMIC HBM bytes it has to move (each operand read once,
 * each result written once).  bench.py builds its roofline table from these records; rocprofv3
 * --kernel-trace --stats of the same command must agree (profiles/).
 *   dram_profile_start(max_records): allocate the event pool, clear the log, switch recording on.
 *   dram_profile_stop(): recording off (the log stays readable).
 *   dram_profile_read(out, max): waits for the recorded events (the ONE entry point of the library
 *     that synchronises), fills out[i].ms, returns the number of records (<= max).
 * Launches beyond max_records are not recorded (dram_profile_dropped() counts them). */
enum {
  DRAM_FAM_CONV_WINO2D = 0, /* fused in-plane Winograd conv, fwd + dgrad          (mfma) */
  DRAM_FAM_WINO_IN,         /* 3-D Winograd input / gradient tile transforms      (hbm)  */
  DRAM_FAM_WINO_GEMM_NN,    /* Winograd-domain GEMMs fwd + dgrad, 1x1x1 GEMMs     (mfma) */
  DRAM_FAM_WINO_OUT,        /* 3-D Winograd output transforms (+ epilogues)       (hbm)  */
  DRAM_FAM_WINO_GEMM_TN,    /* Winograd-domain weight-gradient GEMMs              (mfma) */
  DRAM_FAM_WINO_WGRAD_OUT,  /* slab sum + G^T dU G                                (hbm)  */
  DRAM_FAM_WEIGHT_PACK,     /* weight repacking / weight transforms               (hbm)  */
  DRAM_FAM_WGRAD_W2D,       /* in-plane Winograd weight gradient (+ reduce)       (mfma) */
  DRAM_FAM_CONV_IGEMM,      /* direct implicit-GEMM conv, fwd + dgrad             (mfma) */
  DRAM_FAM_CONV_WGRAD,      /* direct weight gradient (+ reduce)                  (mfma) */
  DRAM_FAM_STEM,            /* 7x7x7 stem conv fwd + wgrad                        (mfma) */
  DRAM_FAM_BN,              /* BN apply / backward / statistics folds / add       (hbm)  */
  DRAM_FAM_POOL_UP,         /* max-pool, upsample+concat, up-projection           (hbm)  */
  DRAM_FAM_HEAD_LOSS,       /* heads, dRAM losses                                 (hbm)  */
  DRAM_FAM_OPTIM,           /* fused Adam / SGD                                   (hbm)  */
  DRAM_FAM_PREP,            /* input transforms                                   (hbm)  */
  DRAM_FAM_CONV_BF16,       /* bf16-storage direct conv, fwd + dgrad (bf16 MFMA)  (mfma) */
  DRAM_FAM_WGRAD_BF16,      /* bf16-storage weight gradient (+ reduce)            (mfma) */
  DRAM_FAM_COUNT
};
typedef struct DramProfRecord {
  int32_t family;      /* DRAM_FAM_* */
  int32_t variant;     /* kernel variant within the family (template instance id; informational) */
  double mfma_flops;   /* executed MFMA FLOPs of this launch (0 for non-matrix kernels) */
  double alg_flops;    /* direct-convolution FLOPs (2*M*N*K) the launch stands for; for the 3-D Winograd
                          pipeline they are booked on the GEMM launch; 0 elsewhere */
  double hbm_bytes;    /* algorithmic HBM bytes of this launch */
  float ms;            /* hipEventElapsedTime(start, end) */
  float pad_;
} DramProfRecord;
const char* dram_profile_family_name(int family);
/* 1 when the family's roof is the matrix pipe, 0 when it is HBM bandwidth */
int dram_profile_family_is_mfma(int family);
int dram_profile_start(int max_records);
int dram_profile_stop(void);
int dram_profile_dropped(void);
int dram_profile_read(DramProfRecord* out, int max_records);

/* ------------------------------------------------------------------------- */
/* Convolution geometry (isotropic stride / padding / dilation, cubic kernel). */
typedef struct DramConvDesc {
  int32_t B;             /* batch */
  int32_t D, H, W, Cin;  /* input  (forward sense) */
  int32_t Do, Ho, Wo, Cout; /* output (forward sense) */
  int32_t k;             /* kernel edge: 1 or 3 (7 only via dram_stem_*) */
  int32_t stride, pad, dil;
  int32_t flags;         /* DRAM_CONV_* plan hints; 0 = none */
} DramConvDesc;
/* The caller's network hands a layer's rounding error on with little amplification (the BasicBlock ResNets: 22 / 38
 * BatchNorm layers; measured at full size with every 64->64 layer on F(4,3)^3 tiles: outputs 6.0e-5 / 1.9e-4 from the
 * fp64 oracle against a bar of 1e-3).  The plan then lets the cheaper estimate win on every layer; without the flag the
 * fused in-plane kernels (F(2x2): a tenth of the rounding error) keep the layers below the last decoder stage unless
 * the 3-D pipeline is estimated > 15 % faster -- ResNet-50's 54 BatchNorm layers amplify an early error ~100 x.
 * Results under either plan are within the 1e-3 bar; only the plan differs. */
#define DRAM_CONV_ROUNDING_TOLERANT 1
/* Per-call hint for a DATA GRADIENT (dram_wino_conv3d_bwd_data[_bn]): the caller runs weight-gradient kernels on another
 * stream at the same time.  The HBM-bound Winograd-domain GEMMs then keep the form with ONE workgroup per CU (100 KB of
 * LDS, room for the other stream's workgroups beside it) instead of the two-workgroups-per-CU form, which is 8-15 %
 * faster on a device of its own and takes the overlap away when it is not (measured).  Results are
 * bit-identical either way. */
#define DRAM_CONV_BWD_OVERLAPPED 2

/* Repack [Cout][Cin][k^3] -> wf[tap][Cout][Cin] (forward B-operand, K=Cin contiguous)
 * and wb[tap][Cin][Cout] (data-gradient B-operand).  Either output may be NULL. */
int dram_pack_conv_weight(const float* w, float* wf, float* wb, int Cout, int Cin, int taps,
                          dram_stream_t stream);

/* Forward 3x3x3 / 1x1x1 convolution as an LDS-tiled fp32-MFMA implicit GEMM.
 * Replaces nn.Conv3d at med3d.py:91-100 (conv3x3x3), :152-157 (Bottleneck), :67/:76
 * (decoder, with bias), :226/:325 (us3).
 *   x  [B,D,H,W,Cin]   wf packed [k^3][Cout][Cin]   bias [Cout] or NULL
 *   y  [B,Do,Ho,Wo,Cout] (pre-BatchNorm output)
 *   stats_partial: NULL, or [dram_conv_num_mtiles(desc)][2][Cout] -- per-M-tile
 *     sum(y) and sum(y*y) per channel (BatchNorm batch statistics, fused epilogue). */
int dram_conv3d_fwd(const float* x, const float* wf, const float* bias, float* y,
                    float* stats_partial, const DramConvDesc* desc, dram_stream_t stream);

/* Data gradient: dx[B,D,H,W,Cin] = conv^T(dy[B,Do,Ho,Wo,Cout], w); wb packed [k^3][Cin][Cout].
 * Optional fused epilogue: dx += add * (gate > 0)   (gate NULL => dx += add); both
 * shaped like dx.  This is the identity-shortcut gradient of med3d.py:141-142/:181-182.
 * Replaces autograd's convolution_backward (input) for the call sites above. */
int dram_conv3d_bwd_data(const float* dy, const float* wb, float* dx, const float* add,
                         const float* gate, const DramConvDesc* desc, dram_stream_t stream);

/* Weight gradient.  dw is written in the reference layout [Cout][Cin][k^3].
 * workspace: split-K slabs, size from dram_conv3d_bwd_weight_workspace().
 * Deterministic (slab + ordered reduce, no float atomics). */
size_t dram_conv3d_bwd_weight_workspace(const DramConvDesc* desc);
int dram_conv3d_bwd_weight(const float* x, const float* dy, float* dw, const DramConvDesc* desc,
                           void* workspace, size_t workspace_bytes, dram_stream_t stream);

/* number of M tiles (rows of stats_partial) dram_conv3d_fwd produces for desc */
int dram_conv_num_mtiles(const DramConvDesc* desc);

/* ------------------------------------------------------------------------- */
/* 3-D Winograd pipeline (F(2,3) or F(4,3) per axis) for stride-1 3x3x3 convolutions with pad == dil and
 * Cin, Cout multiples of 64 (the BasicBlock / Bottleneck conv3x3x3 sites, med3d.py:91-100,
 * and their autograd gradients): 64 instead of 216 multiplies per 2x2x2 output tile and
 * channel pair with F(2,3), 216 instead of 1728 per 4x4x4 tile with F(4,3).  Same tensors and layouts as dram_conv3d_*; the packed weights are the
 * transformed ones and every pass needs a caller-owned workspace.
 *   dram_wino_applicable: 1 when the geometry is supported.
 *   dram_conv_wgrad_algo: plan for the WEIGHT gradient of desc, decided separately from
 *                         dram_conv_algo: 0 direct (dram_conv3d_bwd_weight), 1 this pipeline,
 *                         2 in-plane Winograd z-walking kernel (dram_wgrad_w2d).
 *   dram_wino_num_points: P = Winograd points of desc's tiling (outputs per tile along z, y, x: the
 *                         largest of 4x4x4 / 4x4x2 / 4x2x2 / 2x2x2 whose F(4,3) axes divide the
 *                         dilation sub-lattice extent and that leaves >= 512 tiles):
 *                         216, 144, 96 or 64.
 *   dram_wino_num_points_bwd: Pb = points of the DATA-GRADIENT tiling of desc (the plan may tile that
 *                         pass on its own; forward and weight gradient share the cached V and so
 *                         one tiling).  Today Pb == P.
 *   dram_wino_pack_weight: w [Cout][Cin][27] -> uf [P][Cout][Cin], ub [Pb][Cin][Cout]
 *                         (taps flipped, data-gradient operand); either may be NULL.
 *   dram_wino_workspace(desc, pass): bytes for pass 0 forward, 1 data gradient, 2 weight
 *                         gradient (0 when unsupported).
 *   dram_wino_num_stat_rows: rows of stats_partial written by dram_wino_conv3d_fwd.
 *   v_keep / v_cache: optional [dram_wino_v_elems(desc)] floats; the forward pass leaves the
 *                         transformed input there and the weight gradient reuses it instead of
 *                         transforming x again (x may then be NULL).
 * Deterministic (no atomics; the weight gradient sums its slabs in a fixed order).
 * Arithmetic of the Winograd-domain GEMMs: fp32 MFMA by default.  The environment variable
 * DRAM_MATH = "bf16x3" | "bf16" (opt-in, read per call) switches uf / ub / V and the transformed
 * gradients to a split-bf16 image of the same size (hi = bf16(v), lo = bf16(v - hi) per 32-channel
 * block) and the GEMMs to v_mfma_f32_32x32x16_bf16: three products hi*hi + hi*lo + lo*hi per fp32
 * product ("bf16x3", ~2^-16 per product) or hi*hi alone ("bf16", F(2,3) tiles only).  Weights packed
 * and V cached under one mode must be consumed under the same mode. */
int dram_wino_applicable(const DramConvDesc* desc);
int dram_conv_wgrad_algo(const DramConvDesc* desc);
int dram_wino_num_points(const DramConvDesc* desc);
int dram_wino_num_points_bwd(const DramConvDesc* desc);
int dram_wino_pack_weight(const float* w, float* uf, float* ub, const DramConvDesc* desc, dram_stream_t stream);
size_t dram_wino_workspace(const DramConvDesc* desc, int pass);
int dram_wino_num_stat_rows(const DramConvDesc* desc);
size_t dram_wino_v_elems(const DramConvDesc* desc);
int dram_wino_conv3d_fwd(const float* x, const float* uf, const float* bias, float* y,
                         float* stats_partial, float* v_keep, const DramConvDesc* desc, void* workspace,
                         size_t workspace_bytes, dram_stream_t stream);
/* The same convolution reading the PRE-BatchNorm output x_pre of the producing unit: BatchNorm-apply + ReLU
 * (max(x*pscale + pshift, 0), zero padding outside the volume) run inside the input transform (med3d.py:121-124: bn,
 * relu, next conv) -- the activation tensor in between is never written.  F(4,3)^3 tilings, fp32 math only
 * (dram_wino_prologue_supported). */
int dram_wino_prologue_supported(const DramConvDesc* d);
/* The same forward convolution on an input given as TWO channel blocks, x0 [B,D,H,W,C0] | x1 [B,D,H,W,C1] with
 * C0 + C1 = desc->Cin (both multiples of 64): crop_concat_5d feeding conv_blocks[0] of a decoder block (reference
 * med3d.py:39-48, :87) WITHOUT the concatenated tensor -- each source's input transform writes its channel range of the
 * Winograd-domain image; output, statistics and the kept image are bit-identical to dram_wino_conv3d_fwd on the
 * concatenation.  Where dram_wino_prologue_supported(desc) (F(4,3)^3 tiles). */
int dram_wino_conv3d_fwd_cat(const float* x0, int C0, const float* x1, int C1, const float* uf, const float* bias, float* y,
                             float* stats_partial, float* v_keep, const DramConvDesc* desc, void* workspace,
                             size_t workspace_bytes, dram_stream_t stream);
int dram_wino_conv3d_fwd_bn(const float* x_pre, const float* pscale, const float* pshift, const float* uf, const float* bias,
                            float* y, float* stats_partial, float* v_keep, const DramConvDesc* d, void* workspace,
                            size_t workspace_bytes, dram_stream_t stream);
int dram_wino_conv3d_bwd_data(const float* dy, const float* ub, float* dx, const float* add,
                              const float* gate, const DramConvDesc* desc, void* workspace,
                              size_t workspace_bytes, dram_stream_t stream);
/* The data gradient that also takes the BatchNorm-backward statistics of the unit IN FRONT of this convolution
 * (med3d.py:121-124 / :153-161 backward: conv <- relu <- bn): dx is that unit's dz, and with its pre-BatchNorm output
 * bn_y [B,D,H,W,Cin], its batch mean / invstd and the fused scale / shift of its forward pass the output transform
 * writes, per tile block, the rows (sum g, sum g * xhat) with g = dx * (bn_y*scale + shift > 0),
 * xhat = (bn_y - mean) * invstd -- what dram_bn_bwd_reduce (relu = 1, z = NULL) produces in a pass of its own over dx
 * and bn_y.  stats_partial: [dram_wino_num_stat_rows_bwd(desc)][2][Cin] floats, to be folded with
 * dram_fold_partials.  dx is bit-identical to dram_wino_conv3d_bwd_data(add = gate = NULL). */
int dram_wino_num_stat_rows_bwd(const DramConvDesc* desc);
int dram_wino_conv3d_bwd_data_bn(const float* dy, const float* ub, float* dx, const float* bn_y, const float* bn_mean,
                                 const float* bn_invstd, const float* bn_scale, const float* bn_shift,
                                 float* stats_partial, const DramConvDesc* desc, void* workspace, size_t workspace_bytes,
                                 dram_stream_t stream);
int dram_wino_conv3d_bwd_weight(const float* x, const float* v_cache, const float* dy, float* dw,
                                const DramConvDesc* desc, void* workspace, size_t workspace_bytes,
                                dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Fused in-plane Winograd F(2x2, 3x3) x direct-z kernel for narrow (<= 128-channel) stride-1
 * 3x3x3 convolutions with pad == dil == 1, Cin and Cout multiples of 32 (layer1 and decoder
 * conv sites, med3d.py:91-100, :67/:76): forward and data gradient in one kernel each, 2.25x
 * fewer MFMA products than the direct implicit GEMM, nothing extra in HBM.  Weight gradient
 * stays on dram_conv3d_bwd_weight.
 *   dram_wino2d_pack_weight: w [Cout][Cin][27] -> uf [16][3][Cout][Cin], ub [16][3][Cin][Cout]
 *                         (taps flipped); either may be NULL.
 *   stats_partial rows: dram_wino2d_num_stat_rows(desc). */
int dram_wino2d_applicable(const DramConvDesc* desc);
int dram_wino2d_num_stat_rows(const DramConvDesc* desc);
int dram_wino2d_pack_weight(const float* w, float* uf, float* ub, int Cout, int Cin, dram_stream_t stream);
int dram_wino2d_conv3d_fwd(const float* x, const float* uf, const float* bias, float* y,
                           float* stats_partial, const DramConvDesc* desc, dram_stream_t stream);
int dram_wino2d_conv3d_bwd_data(const float* dy, const float* ub, float* dx, const float* add,
                                const float* gate, const DramConvDesc* desc, dram_stream_t stream);

/* Weight gradient of the same narrow layers through the in-plane Winograd domain (z-walking,
 * 48 instead of 108 MFMA products per 2x2 tile); dw in the reference layout [Cout][Cin][27];
 * workspace = column slabs, summed in a fixed order (deterministic). */
int dram_wgrad_w2d_applicable(const DramConvDesc* desc);
size_t dram_wgrad_w2d_workspace(const DramConvDesc* desc);
int dram_wgrad_w2d(const float* x, const float* dy, float* dw, const DramConvDesc* desc, void* workspace,
                   size_t workspace_bytes, dram_stream_t stream);

/* 1x1x1 stride-1 convolutions (Bottleneck conv1 / conv3, med3d.py:152-157) as plain GEMMs on the
 * batched-GEMM kernels of the Winograd pipeline (M = voxels, a multiple of 256; Cin, Cout multiples of
 * 64), with the same fused epilogues as dram_conv3d_*.  w2d is the reference weight [Cout][Cin][1][1][1]
 * as it is; wt its transpose [Cin][Cout] (dram_pack_conv_weight's wb with taps = 1).
 * stats_partial rows: dram_conv1x1_num_stat_rows(desc).  Weight gradient: split over voxels into slabs
 * (workspace), summed in a fixed order. */
int dram_conv1x1_applicable(const DramConvDesc* desc);
int dram_conv1x1_num_stat_rows(const DramConvDesc* desc);
int dram_conv1x1_fwd(const float* x, const float* w2d, const float* bias, float* y, float* stats_partial,
                     const DramConvDesc* desc, dram_stream_t stream);
int dram_conv1x1_bwd_data(const float* dy, const float* wt, float* dx, const float* add, const float* gate,
                          const DramConvDesc* desc, dram_stream_t stream);
size_t dram_conv1x1_bwd_weight_workspace(const DramConvDesc* desc);
int dram_conv1x1_bwd_weight(const float* x, const float* dy, float* dw, const DramConvDesc* desc, void* workspace,
                            size_t workspace_bytes, dram_stream_t stream);

/* The library's plan for one convolution: 0 direct implicit GEMM (dram_conv3d_*), 1 Winograd
 * F(2x2x2,3x3x3) pipeline (dram_wino_*), 2 fused in-plane Winograd (dram_wino2d_*), 3 1x1x1 GEMM
 * (dram_conv1x1_*; dram_conv_wgrad_algo likewise).
 * env DRAM_CONV_ALGO: 1 = always direct, 2 / 3 = path 1 / 2 wherever applicable (tests). */
int dram_conv_algo(const DramConvDesc* desc);

/* ------------------------------------------------------------------------- */
/* Stem: Conv3d(1,64,k=7,s=2,p=3,bias=False)  (med3d.py:196-202 / :296-302).
 *   x [B,D,H,W] (C=1), w [64][1][7][7][7], y [B,Do,Ho,Wo,64], Do=(D+6-7)/2+1 ...
 *   stats_partial [dram_stem_num_tiles][2][64] or NULL. */
int dram_stem_num_tiles(int B, int Do, int Ho, int Wo);
int dram_stem_fwd(const float* x, const float* w, float* y, float* stats_partial, int B, int D, int H,
                  int W, dram_stream_t stream);
size_t dram_stem_bwd_weight_workspace(int B, int D, int H, int W);
int dram_stem_bwd_weight(const float* x, const float* dy, float* dw, int B, int D, int H, int W,
                         void* workspace, size_t workspace_bytes, dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* BatchNorm3d (+ReLU, + residual) -- med3d.py:121-124,133-142,153-182,203-204,227-228.
 *
 * dram_reduce_partials: sums[r][c] (double) = sum_p partial[p][r][c], r < R.
 *   Used for the BN statistics, BN-backward sums and bias gradients.  In DDP the
 *   caller all-reduces `sums` (SyncBatchNorm, train.py:101) before finalising.
 *   Two-stage for many partials: scratch = dram_reduce_partials_stages(nparts)*R*C doubles
 *   (may be NULL when stages == 1).  Fixed summation order (deterministic).
 *   has_tail != 0: also writes sums[R*C] = tail (the caller's buffer then has R*C+1 doubles): the
 *   rank's element count rides behind the sums so that ONE all-reduce gives global sums AND the
 *   global count (ranks may hold different batch sizes -- torch SyncBatchNorm semantics). */
int dram_reduce_partials_stages(int nparts);
/* The same fold in ONE launch (S = dram_fold_partials_stages(nparts) stage rows; for S > 1 the last block of a column
 * group to finish folds them, fixed order): scratch = S*R*C doubles, followed for S > 1 by DRAM_FOLD_TICKET_DOUBLES
 * doubles of per-call ticket words which the call zeroes itself (a memset on `stream`: launches in flight together on
 * other streams, or a graph replay beside an eager step, never share a counter); sums_f32: optional float copy of the sums
 * ([R][C]; with f32_row1 != NULL and R == 2, row 1 goes there instead: two separately allocated [C] tensors).
 * dram_bn_fold_finalize: fold of the [nparts][2][C] convolution-epilogue partials + dram_bn_finalize on the result
 * (host-side count) in that launch -- the single-process training forward. */
#define DRAM_FOLD_TICKET_DOUBLES 256
int dram_fold_partials_stages(int nparts);
int dram_fold_partials(const float* partial, double* sums, double* scratch, float* sums_f32, float* f32_row1, int nparts,
                       int R, int C, double tail, int has_tail, dram_stream_t stream);
int dram_bn_fold_finalize(const float* partial, double* sums, double* scratch, int nparts, int C, double count,
                          const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                          float eps, int update_running, float* mean, float* invstd, float* scale, float* shift,
                          dram_stream_t stream);
int dram_reduce_partials(const float* partial, double* sums, double* scratch, int nparts, int R, int C,
                         double tail, int has_tail, dram_stream_t stream);

/* training: mean/var from sums[2][C] over `count` elements per channel;
 * eval (sums == NULL): running stats.  Writes mean, invstd, scale = gamma*invstd,
 * shift = beta - mean*scale; when update_running != 0 also
 * running = (1-momentum)*running + momentum*{mean, unbiased var}. */
/* count_dev (device pointer, may be NULL): when given the element count is read from device memory
 * (the all-reduced tail of dram_reduce_partials) and `count` is ignored -- no host round trip. */
int dram_bn_finalize(const double* sums, double count, const double* count_dev, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps,
                     int update_running, float* mean, float* invstd, float* scale, float* shift,
                     int C, dram_stream_t stream);

/* z = act(y*scale[c] + shift[c] + residual).  residual: NULL, or a tensor
 * [B,Dr,Hr,Wr,Cr] sampled at (z*rs, y*rs, x*rs) with channels >= Cr reading 0 --
 * identity shortcut (rs=1, same shape) or shortcut type A (med3d.py:103-112, strided
 * subsample + zero channel pad).  relu != 0 => max(0,.) */
int dram_bn_apply(const float* y, const float* scale, const float* shift, const float* residual,
                  int Dr, int Hr, int Wr, int Cr, int rs, float* z, int B, int D, int H, int W, int C,
                  int relu, dram_stream_t stream);

/* Backward, phase 1: g = dz * (z > 0) (relu != 0) ; partial[p][0][c] = sum g,
 * partial[p][1][c] = sum g * xhat, xhat = (y-mean)*invstd.  nparts from dram_colsum_nparts.
 * The ReLU mask comes from the saved output z, or -- z == NULL, for a BatchNorm without residual --
 * is re-derived from y as fma(y, scale, shift) > 0 with the forward's scale / shift vectors
 * (bitwise the forward's decision; saves one tensor read in each phase). */
int dram_colsum_nparts(long long rows, int C);
int dram_bn_bwd_reduce(const float* dz, const float* z, const float* y, const float* mean,
                       const float* invstd, const float* scale, const float* shift, float* partial,
                       long long rows, int C, int relu, dram_stream_t stream);
/* phase 2: dy = gamma*invstd*(g - sums[0]/count - xhat*sums[1]/count).
 * colsum_partial: NULL, or [dram_bn_bwd_apply_nparts(rows, C)][C] -- per-workgroup column sums of dy (the
 * bias gradient of the convolution in front, med3d.py:67/:76/:226, without a pass of its own); nparts < 1
 * => unsupported for this C (C / 4 must divide 256), use dram_colsum.  count_dev: as in dram_bn_finalize. */
int dram_bn_bwd_apply_nparts(long long rows, int C);
int dram_bn_bwd_apply(const float* dz, const float* z, const float* y, const float* mean,
                      const float* invstd, const float* gamma, const float* scale, const float* shift,
                      const double* sums, double count, const double* count_dev, float* dy,
                      float* colsum_partial, long long rows, int C, int relu, dram_stream_t stream);
/* partial[p][0][c] = sum_rows a[row][c]  (conv-bias gradient) */
int dram_colsum(const float* a, float* partial, long long rows, int C, dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* MaxPool3d(k=3,s=2,p=1) -- med3d.py:206/:275.  argmax: uint8 tap index (first max
 * wins, scan order kd,kh,kw like ATen).  Backward is a gather (deterministic);
 * dx = maxpool^T(dy) (+ add if not NULL).  add: voxel-major like dx with add_stride floats per voxel
 * (add_stride == C: a tensor shaped like dx; larger: a channel slice of a wider tensor, e.g. the skip half
 * of the decoder's concat gradient read in place). */
int dram_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int B, int D, int H, int W, int C,
                     dram_stream_t stream);
/* Stem: BatchNorm-apply + ReLU + max-pool in one pass over the pre-BN tensor y (med3d.py:272-275): writes z =
 * relu(y*scale + shift) (full resolution: skip connection, ReLU mask), the pooled tensor and the taps -- bit-identical to
 * dram_bn_apply followed by dram_maxpool_fwd, one read of the tensor less. */
int dram_bn_maxpool_fwd(const float* y, const float* scale, const float* shift, float* z, float* pooled, uint8_t* argmax,
                        int B, int D, int H, int W, int C, dram_stream_t stream);
int dram_maxpool_bwd(const float* dy, const uint8_t* argmax, const float* add, int add_stride, float* dx,
                     int B, int D, int H, int W, int C, dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* dRAM up-projection inside the decoder: nn.Upsample(x2, trilinear, align_corners=True)
 * + crop_concat_5d -- med3d.py:83-87, :39-48.
 *   src  [B,Ds,Hs,Ws,Cu]  -> upsampled to (2Ds,2Hs,2Ws)
 *   skip [B,Dk,Hk,Wk,Ck]  centre-cropped to the upsampled size
 *   cat  [B,2Ds,2Hs,2Ws,Cu+Ck]  (upsampled channels FIRST)
 * skip == NULL with Ck == 0 (Dk, Hk, Wk >= the up-sampled extents): the up-sampled tensor alone -- the convolution
 * behind it takes the skip tensor as a second source (dram_wino_conv3d_fwd_cat) and `cat` is never built. */
int dram_upcat_fwd(const float* src, const float* skip, float* cat, int B, int Ds, int Hs, int Ws, int Cu,
                   int Dk, int Hk, int Wk, int Ck, dram_stream_t stream);
/* dsrc = upsample^T(dcat[..., :Cu]);  dskip (full skip shape, zero outside the crop)
 * = dcat[..., Cu:].  Either output may be NULL. */
int dram_upcat_bwd(const float* dcat, float* dsrc, float* dskip, int B, int Ds, int Hs, int Ws, int Cu,
                   int Dk, int Hk, int Wk, int Ck, dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* First decoder convolution of us1 WITHOUT the up-sampled tensor (csrc/upmix.hip) -- med3d.py:83-87 with
 * conv_blocks[0] (med3d.py:67): conv(concat(up(a), s)) = sum_t up(W_up[t] . a) shifted by tap t + conv_s(s).
 * The channel mixing W_up[t] . a is a plain GEMM at the LOW resolution (the 1x1x1 entry points, Cout = 27 Co);
 * what these entry points add is the tap/trilinear gather, separable per axis, and the weight split / merge.
 *
 * dram_upmix_axis_fwd:  In[outer][Ni][m][3][C] (float, or bf16 when in_bf16) -> Out[outer][2 Ni][m][C] (float):
 *   Out[o][v][m][c] = sum_k [0 <= u = v+k-1 < 2Ni] (w0(u) In[o][i0(u)][m][k][c] + w1(u) In[o][i1(u)][m][k][c]),
 *   (i0, i1, w0, w1)(u) = PyTorch's align_corners source index of destination u, scale (Ni-1)/(2Ni-1).
 *   Three calls: X (outer = B Ds Hs, Ni = Ws, m = 9), Y (outer = B Ds, Ni = Hs, m = 2Ws 3), Z = ..._final.
 * dram_upmix_axis_fwd_final: the same pass, + base (NULL, or the skip convolution's output; may alias out), stored
 *   as float / bf16, with BatchNorm partial sums of the stored values: stats [dram_upmix_stat_rows(nvox)][2][C]
 *   (NULL: none), nvox = outer 2Ni m, C <= 128.
 * dram_upmix_axis_bwd: the exact transpose, G[outer][2Ni][m][C] -> H[outer][Ni][m][3][C] (gather form, no atomics).
 * dram_upmix_split_weight: w [Co][Cu+Cs][27] -> wlo [27 Co][Cu] (row t Co + co) and ws [Co][Cs][27];
 * dram_upmix_merge_wgrad: the inverse, for the two weight gradients. */
int dram_upmix_stat_rows(long long nvox);
int dram_upmix_axis_fwd(const void* in, int in_bf16, float* out, long long outer, int Ni, int m, int C,
                        dram_stream_t stream);
int dram_upmix_axis_fwd_final(const float* in, const void* base, void* out, int out_bf16, float* stats, long long outer,
                              int Ni, int m, int C, dram_stream_t stream);
int dram_upmix_axis_bwd(const void* g, int g_bf16, void* h, int h_bf16, long long outer, int Ni, int m, int C,
                        dram_stream_t stream);
int dram_upmix_split_weight(const float* w, float* wlo, float* ws, int Co, int Cu, int Cs, dram_stream_t stream);
int dram_upmix_merge_wgrad(const float* dwlo, const float* dws, float* dw, int Co, int Cu, int Cs, dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Heads -- med3d.py:283-284 (cls) / :382-387 (reg).
 *   x [B,D,H,W,32]; w [NO][32], bias [NO]  (fcs.0 and fcs.1 stacked: NO = 6+3 or 1+1)
 *   dense [B][NO][D*H*W]  (planar == the reference's NCDHW per head)
 *   sigmoid != 0: dense = sigmoid(.) and pooled sums are lung-weighted:
 *     lungs: NULL (=> ones) or the FULL-RES mask [B,Dl,Hl,Wl], sampled nearest
 *     (F.interpolate(mode='nearest'), med3d.py:386).
 *   partial [B][nblk][NO+1]: per-block sum(dense*L) per output, and sum(L) last. */
int dram_head_nblk(long long voxels_per_sample);
int dram_head_fwd(const float* x, const float* w, const float* bias, const float* lungs, int Dl, int Hl,
                  int Wl, float* dense, float* partial, int B, int D, int H, int W, int NO, int sigmoid,
                  dram_stream_t stream);
/* backward: dpre[c] = (gdense[b][c][v] + gpool[b][c]*L[v]) * (sigmoid ? s(1-s) : 1)
 *   gdense may be NULL.  dx [B,D,H,W,32] = sum_c w[c][:] dpre[c];
 *   wpartial [B*nparts][NO][33]: per-block sum dpre[c]*x[k] (k<32) and sum dpre[c] (k=32);
 *   nparts = dram_head_bwd_nparts(D*H*W) blocks per sample. */
int dram_head_bwd_nparts(long long voxels_per_sample);
int dram_head_bwd(const float* x, const float* w, const float* dense, const float* gdense,
                  const float* gpool, const float* lungs, int Dl, int Hl, int Wl, float* dx,
                  float* wpartial, int B, int D, int H, int W, int NO, int sigmoid, dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* dRAM segmentation losses -- models.py:523-531 + metrics.py:10-37, label prep
 * models.py:567-570.  cle/pse: [B,1,D,H,W] dense maps; lungs/ems: full-res masks
 * [B,Dl,Hl,Wl] sampled nearest; binary[b] in {0,1} multiplies ems; smoothness = the
 * in-mask weight of metrics.py:10,24 (0.85 at the models.py:529 call site).
 *   partial [nblk][6]: sum t, A1, A0, I, S1, S2 (see csrc/loss.hip). */
int dram_segloss_nblk(long long voxels_total);
int dram_segloss_fwd(const float* cle, const float* pse, const float* lungs, const float* ems,
                     const float* binary, int Dl, int Hl, int Wl, float* partial, int B, int D, int H,
                     int W, float smoothness, dram_stream_t stream);
/* coef[8] (device): c_dice_a, c_dice_b, c_bce1, c_bce0, -, -, -, - (see loss.hip) */
int dram_segloss_bwd(const float* cle, const float* pse, const float* lungs, const float* ems,
                     const float* binary, int Dl, int Hl, int Wl, const float* coef, float* gcle,
                     float* gpse, int B, int D, int H, int W, float smoothness, dram_stream_t stream);

/* The O(B) tail of the regression train loss (models.py:549-574) in ONE launch: the fp64 fold of the seg-loss
 * block sums, dice + balanced BCE (metrics.py:10-37), the two interval losses (models.py:512-521) with the band
 * of each label looked up in *_bands [n][2] (models.py:492-510), loss = cle + pse + 2 mul + seg.
 *   out [5]: loss, loss_cle, loss_pse, mul_loss, seg_loss;  coef [8]: dram_segloss_bwd's input for d loss;
 *   greg [2][B]: d loss / d reg_outs.  A label outside its band table makes the loss NaN. */
int dram_regloss_tail(const float* partial, int nblk, const float* reg_cle, const float* reg_pse,
                      const long long* cle_labels, const long long* pse_labels, const float* cle_w,
                      const float* pse_w, const float* cle_bands, int n_cle, const float* pse_bands, int n_pse,
                      int B, double voxels_total, double smooth, double beta, double gamma, float* out,
                      float* coef, float* greg, dram_stream_t stream);

/* Predict-time up-projection (models.py:438-441): out = trilinear(dense -> Do,Ho,Wo,
 * align_corners) * ess; partial [B][nblk] per-block sums of out. */
int dram_upproject_nblk(long long voxels_per_sample);
int dram_upproject(const float* dense, const float* ess, float* out, float* partial, int B, int D, int H,
                   int W, int Do, int Ho, int Wo, dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Multi-tensor optimizers -- torch.optim.Adam at models.py:385-387 / :689-691 and the
 * SGD arguments of train.py:25,27.  table: device array of DramTensorRef[ntensors];
 * chunks: device array of DramChunkRef[nchunks] (tensor index + element offset). */
typedef struct DramTensorRef {
  float* p;
  const float* g;
  float* m;
  float* v;
  int64_t n;
} DramTensorRef;
typedef struct DramChunkRef {
  int32_t tensor;
  int32_t pad;
  int64_t offset;
} DramChunkRef;
#define DRAM_OPT_CHUNK 16384
/* Work list of dram_pack_conv_weight_bf16_multi: weight [Cout][Cin][taps] (fp32) -> its bf16 packed copies wf
 * [taps][Cout][Cin] / wb [taps reversed][Cin][Cout] at ELEMENT offsets off_f / off_b of one flat bf16 buffer (-1: not
 * wanted); chunks enumerate the tiles of each weight: offset = 0 .. dram_pack_conv_weight_bf16_tiles(...) - 1. */
typedef struct DramPackRef {
  const float* w;
  int64_t off_f;
  int64_t off_b;
  int32_t Cout, Cin, taps, pad;
} DramPackRef;
int dram_adam_multi(const DramTensorRef* table, const DramChunkRef* chunks, int nchunks, float lr,
                    float beta1, float beta2, float eps, float weight_decay, float bias_corr1,
                    float bias_corr2, float grad_scale, dram_stream_t stream);
/* Same update with every hyper-parameter and the step count in DEVICE memory:
 *   hyper = float[7] {lr, beta1, beta2, eps, weight_decay, grad_scale, step}
 * step is incremented on the device first and the bias corrections 1 - beta^step are computed in the
 * kernel (double), so a hipGraph captured around a train step stays valid across steps and lr changes
 * (the host only rewrites hyper[0] when the scheduler moves lr). */
int dram_adam_multi_dev(const DramTensorRef* table, const DramChunkRef* chunks, int nchunks, float* hyper,
                        dram_stream_t stream);
int dram_sgd_multi(const DramTensorRef* table, const DramChunkRef* chunks, int nchunks, float lr,
                   float momentum, float weight_decay, int first_step, float grad_scale,
                   dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* Deterministic input transforms of the reference data module (models.py:59-63):
 * IntensityWindow (functional.py:13-26), Standardize (intensity_transforms.py:108-111),
 * Interpolate(align_corners, only_in_plane) for images / masks (spatial_transforms.py:55-98).
 *   scan [D,H,W] raw HU as float32; zidx [Do] int32 = torch.linspace(0, D-1, Do).long();
 *   dram_window_stats -> partial [nblk][2] (sum w, sum w^2 of the windowed values);
 *   mean_invstd [2] (device) = {mean, 1/std(unbiased)} of the windowed volume;
 *   dram_prep_image -> out [Do,Ho,Wo]; dram_prep_mask -> nearest-resized mask. */
int dram_window_stats_nblk(long long n);
int dram_window_stats(const float* scan, float* partial, long long n, float lo, float hi, dram_stream_t stream);
int dram_prep_image(const float* scan, const int* zidx, const float* mean_invstd, float* out, int D, int H, int W,
                    int Do, int Ho, int Wo, float lo, float hi, dram_stream_t stream);
int dram_prep_mask(const float* mask, const int* zidx, float* out, int D, int H, int W, int Do, int Ho, int Wo,
                   dram_stream_t stream);

/* Predict post-processing (processor.py:111-129, :143): src [D,H,W] resized (trilinear, align_corners=True) to
 * the lung-crop size (rd,rh,rw) and pasted at offset (oz,oy,ox) into a zero volume of the original grid
 * [Do,Ho,Wo]; out_f32 and/or out_u8 (= utils.windowing(., (0,1) -> (0,255)) truncated to uint8). */
int dram_resample_paste(const float* src, float* out_f32, uint8_t* out_u8, int D, int H, int W, int rd, int rh, int rw,
                        int oz, int oy, int ox, int Do, int Ho, int Wo, dram_stream_t stream);

/* Train-time augmentations (models.py:66-74) with GIVEN parameters, fused into one gather pass:
 * GaussianAddictive (intensity_transforms.py:145-177; noise [D,H,W] supplied by the caller, minmax[2] = volume
 * {min, max} on the device, e.g. folded from dram_minmax partials [nblk][2]), BoxMaskOut (:180-237), Flip
 * (spatial_transforms.py:100-131), CropAndResize (:133-197, functional.py:68-94: affine_grid + grid_sample;
 * image trilinear / align_corners=True, mask nearest / align_corners=False, zero padding).
 * flags: bit 0 noise, 1 boxes, 2 flip, 3 crop-resize; boxes[b] = {z0,z1,y0,y1,x0,x1} half-open, pre-flip grid;
 * flip_axes: bit 0 z, 1 y, 2 x; box_lo/box_hi: bounding box / size per axis (z, y, x).  x != out. */
typedef struct DramAugment {
  int32_t flags;
  int32_t n_boxes;
  int32_t boxes[10][6];
  int32_t flip_axes;
  float sigma;
  float box_lo[3];
  float box_hi[3];
} DramAugment;
int dram_minmax_nblk(long long n);
int dram_minmax(const float* x, float* partial, long long n, dram_stream_t stream);
int dram_augment_image(const float* x, const float* noise, const float* minmax, float* out, int D, int H, int W,
                       const DramAugment* aug, dram_stream_t stream);
int dram_augment_mask(const float* mask, float* out, int D, int H, int W, const DramAugment* aug,
                      dram_stream_t stream);

/* out[i] = a[i] + b[i]  (gradient accumulation where two consumers meet) */
int dram_add(const float* a, const float* b, float* out, long long n, dram_stream_t stream);

/* ------------------------------------------------------------------------- */
/* bf16 STORAGE path -- the reference's `--precision bf16` (train.py:46: Lightning wraps the step in
 * torch.autocast(bfloat16); BASELINE configs[2], [4]).  Activation-sized tensors (x, y, z, dy, dx, residuals,
 * concat buffers) are bf16 NDHWC (`void*`, 2 bytes per element, same offsets); everything else keeps the types of
 * the fp32 entry points: BatchNorm statistics / scale / shift, biases, the dense head outputs and every weight
 * GRADIENT are fp32, statistic folds are double.  Arithmetic: products of bf16 operands accumulated in fp32
 * (v_mfma_f32_32x32x16_bf16), element-wise math in fp32, one rounding (to nearest even) on store.
 * The *_bf16 element-wise entry points take exactly the arguments of their fp32 namesakes. */
long long dram_pack_conv_weight_bf16_tiles(int Cout, int Cin, int taps);
int dram_pack_conv_weight_bf16_multi(const DramPackRef* table, const DramChunkRef* chunks, int nchunks, void* flat,
                                     double total_elems, dram_stream_t stream);
int dram_cast_f32_to_bf16(const float* src, void* dst, long long n, dram_stream_t stream);
int dram_cast_bf16_to_f32(const void* src, float* dst, long long n, dram_stream_t stream);
/* The stride-2 3x3x3 convolution (k 3, stride 2, pad 1, even extents, Cin % 8 == 0) as a stride-1 convolution of the
 * space-to-depth tensor: x8[b][z][y][x][p][c] = x[b][2z + pz][2y + py][2x + px][c], p = 4 pz + 2 py + px (8 Cin
 * channels at half the extents), with the embedded weights w3 [Cout][8 Cin][3][3][3] (dram_s2_embed_weight: 27 of
 * 216 (parity, offset) slots per (co, ci) carry a tap, the rest are zero).  dram_d2s_bf16 is the inverse permutation
 * with the data gradient's optional `+= add * (gate > 0)`; dram_s2_extract_wgrad picks the 27 taps out of the
 * gradient of w3.  The host runs forward / data gradient / weight gradient of the derived stride-1 geometry on the
 * bf16 convolution entry points below. */
int dram_s2d_bf16(const void* x, void* x8, int B, int D, int H, int W, int C, dram_stream_t stream);
int dram_d2s_bf16(const void* dx8, const void* add, const void* gate, void* dx, int B, int D, int H, int W, int C,
                  dram_stream_t stream);
int dram_s2_embed_weight(const float* w, float* w3, int Cout, int Cin, dram_stream_t stream);
int dram_s2_extract_wgrad(const float* dw3, float* dw, int Cout, int Cin, dram_stream_t stream);
/* 1 when the bf16 direct kernels take this geometry: 3x3x3, stride 1, pad == dilation, Cin % 32 == Cout % 32 == 0
 * (everything else is run by the host on the fp32 kernels around cast passes) */
int dram_conv_bf16_supported(const DramConvDesc* d);
int dram_conv_bf16_num_stat_rows(const DramConvDesc* d);
/* w [Cout][Cin][k^3] fp32 -> wf [taps][Cout][Cin] bf16 (forward operand), wb [taps][Cin][Cout] bf16 with the taps
 * flipped (data-gradient operand); either may be NULL */
int dram_pack_conv_weight_bf16(const float* w, void* wf, void* wb, int Cout, int Cin, int taps, dram_stream_t stream);
/* nn.Conv3d forward (med3d.py:11,121,123,...) / its data gradient / its weight gradient on bf16 tensors.
 * stats_partial [dram_conv_bf16_num_stat_rows][2][Cout] fp32: per-tile sums of y and y^2 (of the ROUNDED values the
 * BatchNorm pass reads back), or NULL.  bwd_data: dx = conv_transpose(dy) (+ add * (gate > 0), both bf16, optional). */
int dram_conv3d_fwd_bf16(const void* x, const void* wf, const float* bias, void* y, float* stats_partial,
                         const DramConvDesc* d, dram_stream_t stream);
int dram_conv3d_bwd_data_bf16(const void* dy, const void* wb, void* dx, const void* add, const void* gate,
                              const DramConvDesc* d, dram_stream_t stream);
size_t dram_conv3d_bwd_weight_bf16_workspace(const DramConvDesc* d);
int dram_conv3d_bwd_weight_bf16(const void* x, const void* dy, float* dw, const DramConvDesc* d, void* workspace,
                                size_t workspace_bytes, dram_stream_t stream);
int dram_stem_fwd_bf16(const float* x, const float* w, void* y, float* stats_partial, int B, int D, int H, int W,
                       dram_stream_t stream);
int dram_stem_bwd_weight_bf16(const float* x, const void* dy, float* dw, int B, int D, int H, int W, void* workspace,
                              size_t workspace_bytes, dram_stream_t stream);
/* the same two on the bf16 matrix cores (input rounded to bf16 on its way into LDS, weights rounded per launch;
 * same tile geometry, statistic rows and workspace as the fp32-MFMA forms above) */
int dram_stem_fwd_bf16mm(const float* x, const float* w, void* y, float* stats_partial, int B, int D, int H, int W,
                         dram_stream_t stream);
int dram_stem_bwd_weight_bf16mm(const float* x, const void* dy, float* dw, int B, int D, int H, int W, void* workspace,
                                size_t workspace_bytes, dram_stream_t stream);
int dram_bn_apply_bf16(const void* y, const float* scale, const float* shift, const void* residual, int Dr, int Hr,
                       int Wr, int Cr, int rs, void* z, int B, int D, int H, int W, int C, int relu,
                       dram_stream_t stream);
int dram_bn_bwd_reduce_bf16(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                            const float* scale, const float* shift, float* partial, long long rows, int C, int relu,
                            dram_stream_t stream);
int dram_bn_bwd_apply_bf16(const void* dz, const void* z, const void* y, const float* mean, const float* invstd,
                           const float* gamma, const float* scale, const float* shift, const double* sums, double count,
                           const double* count_dev, void* dy, float* colsum_partial, long long rows, int C, int relu,
                           dram_stream_t stream);
int dram_colsum_bf16(const void* a, float* partial, long long rows, int C, dram_stream_t stream);
int dram_maxpool_fwd_bf16(const void* x, void* y, uint8_t* argmax, int B, int D, int H, int W, int C,
                          dram_stream_t stream);
int dram_bn_maxpool_fwd_bf16(const void* y, const float* scale, const float* shift, void* z, void* pooled, uint8_t* argmax,
                             int B, int D, int H, int W, int C, dram_stream_t stream);
int dram_maxpool_bwd_bf16(const void* dy, const uint8_t* argmax, const void* add, int add_stride, void* dx, int B, int D,
                          int H, int W, int C, dram_stream_t stream);
int dram_upcat_fwd_bf16(const void* src, const void* skip, void* cat, int B, int Ds, int Hs, int Ws, int Cu, int Dk,
                        int Hk, int Wk, int Ck, dram_stream_t stream);
int dram_upcat_bwd_bf16(const void* dcat, void* dsrc, void* dskip, int B, int Ds, int Hs, int Ws, int Cu, int Dk, int Hk,
                        int Wk, int Ck, dram_stream_t stream);
int dram_head_fwd_bf16(const void* x, const float* w, const float* bias, const float* lungs, int Dl, int Hl, int Wl,
                       float* dense, float* partial, int B, int D, int H, int W, int NO, int sigmoid,
                       dram_stream_t stream);
int dram_head_bwd_bf16(const void* x, const float* w, const float* dense, const float* gdense, const float* gpool,
                       const float* lungs, int Dl, int Hl, int Wl, void* dx, float* wpartial, int B, int D, int H, int W,
                       int NO, int sigmoid, dram_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* DRAM_HIP_H */
