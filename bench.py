"""bench.py -- CT volumes/sec of the Med3D + dRAM train step on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 0|1|2|3|4|5]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward + loss + backward + (N>1: RCCL gradient all-reduce, SyncBN exchanges) +
fused Adam update on one batch of synthetic volumes already resident in HBM
(SURVEY.md §8d recipe).  Default workload = BASELINE.json configs[1]:
conf/med3d18.yaml (resnet18segcls), batch 2 per GPU, 1x128x256x256, fp32.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     -- the kernel family with the largest time share ON THE CONFIG RUN, from the library's own
                  kernel timeline (hipEvent pairs around every kernel launch, on the launch stream, over the
                  same K steps run a second time right after the timed region -- the ~1,400 event records per
                  step cost 4-5 % of a step, so `value` is timed without them; --timeline in puts them inside): `achieved` = EXECUTED MFMA FLOPs / time for a matrix-bound family
                  (a Winograd kernel issues fewer products than the direct convolution it replaces; the
                  ratio is `algorithmic_speedup`, never part of `frac`) or ALGORITHMIC HBM bytes / time
                  for a bandwidth-bound one; `frac` = achieved / peak <= 1.  `families` holds the same
                  numbers for every family (>= 99 % of the GPU time of a step), `whole_step` the
                  step-level executed-MFMA and HBM fractions.
  cpu_baseline -- the CPU oracle (oracle/med3d_oracle.py, torch CPU ops): 1 warm-up + 2 timed train steps
                  of one volume of the same shape, host threads stated (N=1, rank 0 only; runs BEFORE the
                  GPU section)
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # id: (factory, per-GPU batch, (D,H,W), train GFLOP per volume [SURVEY.md §8a], fused-minimum activation
    #      elements fwd+bwd per volume [SURVEY.md §8a], parameters)
    0: ("resnet34segcls", 1, (64, 128, 128), 1259.3, (0.233 + 0.256) * 1e9, 64.79e6),
    1: ("resnet18segcls", 2, (128, 256, 256), 6943.0, (1.561 + 1.747) * 1e9, 34.48e6),
    2: ("resnet18segreg", 2, (128, 256, 256), 6941.6, (1.554 + 1.732) * 1e9, 34.48e6),
    3: ("resnet50segreg", 1, (128, 256, 256), 10313.4, (3.458 + 3.838) * 1e9, 47.86e6),
    # configs[4] geometry (full-resolution volume) in fp32 without activation checkpointing: a capacity /
    # int32-offset check of the kernels at 8x the voxels, not a BASELINE metric (that one asks for bf16)
    4: ("resnet50segreg", 1, (256, 512, 512), 8 * 10313.4, (27.67 + 30.70) * 1e9, 47.86e6),
    # the reference's own default job (reference train.py:21,30,42: med3ddram50, target_size 128x224x288, batch 1):
    # 63/64 of config 3's voxels; S2 = 16x28x36, so the dilated stages run ragged F(4,3) Winograd tiles
    5: ("resnet50segreg", 1, (128, 224, 288), 10313.4 * 63 / 64, (3.458 + 3.838) * 1e9 * 63 / 64, 47.86e6),
}
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters
PEAK_BF16_MFMA_TFLOPS = 2500.0  # same guide: ~2.5 PF dense bf16 (v_mfma_f32_32x32x16_bf16, 32 cycles per SIMD)
PEAK_HBM_GBS = 8000.0           # same guide: HBM3E ~8 TB/s
BF16_CONFIGS = (2, 4)           # BASELINE configs[2] and [4] are specified in bf16

FAMILY_DESC = {
    "conv_wino2d": "conv_wino2d_kernel<NJ,..> (fused in-plane Winograd F(2x2,3x3) x direct-z conv, fwd + dgrad, fp32 MFMA 32x32x2)",
    "wino_in": "wino_in*_kernel (3-D Winograd input / gradient tile transforms)",
    "wino_gemm_nn": "wino_gemm_nn_kernel<NJ> (Winograd-domain batched GEMM fwd + dgrad, 1x1x1 GEMMs; fp32 MFMA 32x32x2)",
    "wino_out": "wino_out_kernel (3-D Winograd output transform + bias / shortcut-gradient / BN-statistics epilogue)",
    "wino_gemm_tn": "wino_gemm_tn_kernel (Winograd-domain weight-gradient GEMM, fp32 MFMA)",
    "wino_wgrad_out": "wino_wgrad_out_kernel / slab_sum (slab sum + G^T dU G)",
    "weight_pack": "weight packing / Winograd weight transforms",
    "conv_wgrad_w2d": "conv_wgrad_w2d_kernel + reduce (in-plane Winograd weight gradient, fp32 MFMA)",
    "conv_igemm": "conv_igemm*_kernel (direct implicit-GEMM conv fwd + dgrad, fp32 MFMA 32x32x2)",
    "conv_wgrad": "conv_wgrad*_kernel + reduce (direct weight gradient, fp32 MFMA)",
    "stem": "stem_fwd_kernel / stem_wgrad_kernel (7x7x7 stride-2 conv, C_in = 1; fp32 MFMA, or bf16 MFMA on the bf16 storage path)",
    "bn_elementwise": "BN statistics folds, BN-apply(+residual+ReLU), BN backward reduce / apply, add",
    "pool_up": "max-pool fwd/bwd, trilinear upsample + concat fwd/bwd, up-projection",
    "head_loss": "1x1x1 heads + pooled scores, dRAM loss kernels",
    "optim": "adam_multi / sgd_multi (fused multi-tensor optimizer)",
    "prep": "input transforms",
    "conv_bf16": "conv3_bf16_kernel<NB,EPI> (bf16-storage direct implicit-GEMM 3x3x3 conv, fwd + dgrad, bf16 MFMA 32x32x16)",
    "wgrad_bf16": "wgrad3_bf16_kernel + reduce (bf16-storage weight gradient, transposed LDS operands, bf16 MFMA 32x32x16)",
}
# family -> key in profiles/r*_pmc_hbm_traffic.json
TRAFFIC_KEY = {"conv_bf16": "conv_bf16", "wgrad_bf16": "wgrad_bf16", "conv_wino2d": "conv_wino2d", "wino_in": "wino_in", "wino_gemm_nn": "wino_gemm_nn",
               "wino_out": "wino_out", "wino_gemm_tn": "wino_gemm_tn", "conv_wgrad_w2d": "conv_wgrad_w2d",
               "conv_igemm": "conv_igemm", "bn_elementwise": "bn_elementwise", "stem": "stem"}


def measured_traffic(config=1):
    """HBM bytes per launch per family from the rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in separate
    runs, guide's gfx950 corrections; tools/profile_round.sh + tools/summarize_profile.py), recorded for
    config 1.  PMC counters cannot be read from inside bench.py, so the file carries the fingerprint of the
    kernel sources it was measured on and is IGNORED (traffic = null) when the sources differ."""
    import glob
    from bodyct_dram_emph_subtype_amd import _build
    sfx = "" if config == 1 else f"_config{config}"
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_hbm_traffic{sfx}.json")))
    for path in reversed(files):
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:  # noqa: BLE001
            continue
        if d.get("source_hash") == _build.source_hash():
            return d, os.path.relpath(path, ROOT)
    return None, None


def synth_batch(B, dims, rank, device):
    """image ~ N(0,1) seed 1234+rank; centred-ellipsoid lung mask with
    semi-axes (0.4D, 0.35H, 0.4W); em = (image < -1) & lung; labels randint seed 4321+rank."""
    D, H, W = dims
    g = torch.Generator(device=device).manual_seed(1234 + rank)
    image = torch.randn(B, 1, D, H, W, device=device, generator=g)
    z = (torch.arange(D, device=device).float() - (D - 1) / 2) / (0.4 * D)
    y = (torch.arange(H, device=device).float() - (H - 1) / 2) / (0.35 * H)
    x = (torch.arange(W, device=device).float() - (W - 1) / 2) / (0.4 * W)
    lung = ((z[:, None, None] ** 2 + y[None, :, None] ** 2 + x[None, None, :] ** 2) <= 1.0).float()
    lung = lung[None, None].expand(B, 1, D, H, W).contiguous()
    em = ((image < -1.0).float() * lung).contiguous()
    gl = torch.Generator().manual_seed(4321 + rank)
    cle = torch.randint(0, 6, (B,), generator=gl).to(device)
    pse = torch.randint(0, 3, (B,), generator=gl).to(device)
    return image, lung, em, cle, pse


def make_step(factory, module, opt, batch):
    from bodyct_dram_emph_subtype_amd.models import cls_train_loss, reg_train_loss
    image, lung, em, cle, pse = batch
    if factory.endswith("cls"):
        cw = torch.full((6,), 1.0 / 6, device=image.device)
        pw = torch.full((3,), 1.0 / 3, device=image.device)

        def step():
            opt.zero_grad(set_to_none=True)
            _, outs = module(image, lung)
            loss, _ = cls_train_loss(outs, cle, pse, cw, pw)     # models.py:253-258 (K16: torch glue)
            loss.backward()
            opt.step()
            return loss
        return step
    B = image.shape[0]
    cwt = torch.full((B,), 1.0 / 6, device=image.device)
    pwt = torch.full((B,), 1.0 / 3, device=image.device)

    def step():
        opt.zero_grad(set_to_none=True)
        dense, outs = module(image, lung)
        loss, _ = reg_train_loss(dense, outs, lung, em, cle, pse, cwt, pwt)
        loss.backward()
        opt.step()
        return loss
    return step


_ENV_SAVED = {}


def single_stream(on: bool):
    """Run the following steps on ONE stream (the engine's A/B switch DRAM_WGRAD_STREAM=0, which counts under
    DRAM_TUNING=1 only), or restore what the environment held before."""
    if on:
        for k, v in (("DRAM_TUNING", "1"), ("DRAM_WGRAD_STREAM", "0")):
            _ENV_SAVED[k] = os.environ.get(k)
            os.environ[k] = v
    else:
        for k, v in _ENV_SAVED.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        _ENV_SAVED.clear()


def host_cpu_share():
    """(CPUs the job owns, threads to run CPU work on).  cgroup v2 `cpu.max` = "<quota> <period>" or "max ..."; without
    a quota the job owns every visible CPU."""
    n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            cores = max(1, min(n, -(-int(quota) // int(period))))
            return cores, max(1, min(n, 2 * cores))
    except (OSError, ValueError):
        pass
    return n, n


def cpu_baseline(factory, dims, B=1):
    """The CPU baseline in a CHILD process (this same file with --cpu-baseline-child, GPU hidden from it): its 32 intra-op
    threads must be gone when the GPU section starts.  Run in-process, the idle OpenMP pool kept the 16 CPUs of the job busy
    for the next seconds and the launcher thread of the eager GPU steps ran at half speed -- host issue time 23.5 instead
    of 10.5 ms per step, the timed 20 steps at 41.1 ms where their own hipEvent median says 36.9 (profiles/r05, first
    bench phase of the final sources)."""
    import subprocess
    env = dict(os.environ)
    env["HIP_VISIBLE_DEVICES"] = ""
    env["CUDA_VISIBLE_DEVICES"] = ""
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", factory, *map(str, dims), str(B)],
                       env=env, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"CPU baseline child failed ({r.returncode}): {r.stderr[-2000:]}")
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])


def _cpu_baseline_impl(factory, dims, B=1):
    """Oracle train steps (fwd + loss + bwd + Adam) on the host cores, at the batch the GPU line runs (so BatchNorm
    sees the same batch): 1 warm-up + 2 timed steps at batch 1, 1 warm-up + 1 timed step at batch >= 2 (BASELINE.md
    §3 plan; ~25 s of CPU work for the headline).  Bounded: dims are capped at 128x256x256, the batch at 2."""
    from oracle import med3d_oracle as orc
    from bodyct_dram_emph_subtype_amd import med3d
    dims = tuple(min(a, b) for a, b in zip(dims, (128, 256, 256)))
    torch.manual_seed(0)
    kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
    m = getattr(med3d, factory)(**kw)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    B = max(1, min(int(B), 2))
    image, lung, em, cle, pse = synth_batch(B, dims, 0, "cpu")
    mom = {n: (torch.zeros_like(sd[n]), torch.zeros_like(sd[n])) for n in names}
    # `cores` = CPUs this job OWNS (cgroup cpu.max quota; the GPU box shows all 128 hardware threads of the host but
    # grants 16 CPUs), `threads` = intra-op threads the oracle ran on (2 x the share: measured fastest,
    # tools/cpu_threads_probe.py -- torch's default of one thread per visible hardware thread oversubscribes 8x)
    cores, threads = host_cpu_share()
    torch.set_num_threads(threads)
    times = []
    nsteps = 3 if B == 1 else 2
    for it in range(nsteps):
        t0 = time.perf_counter()
        leaves = {k: (v.detach().requires_grad_(True) if k in names else v) for k, v in sd.items()}
        dense, outs = orc.forward(leaves, image, lung, factory, train=True)
        if factory.endswith("cls"):
            loss, _ = orc.cls_train_loss(outs, cle, pse, torch.full((6,), 1 / 6), torch.full((3,), 1 / 3))
        else:
            loss, _ = orc.reg_train_loss(dense, outs, lung, em, cle, pse, torch.full((1,), 1 / 6), torch.full((1,), 1 / 3))
        loss.backward()
        with torch.no_grad():
            for n in names:
                p = leaves[n].detach()
                orc.adam_step(p, leaves[n].grad, mom[n][0], mom[n][1], it + 1, 1e-4)
                sd[n] = p
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / (nsteps - 1)
    return {"value": B / dt, "unit": "volumes/sec", "cores": cores, "threads": threads, "kind": "port",
            "sample": f"{factory} train step (oracle, torch CPU ops, fwd+loss+bwd+Adam), batch {B} of "
                      f"1x{dims[0]}x{dims[1]}x{dims[2]} (the GPU line's per-GPU batch, capped at 2): 1 warm-up "
                      f"({times[0]:.1f} s) + {nsteps - 1} timed step(s) ({', '.join(f'{t:.1f} s' for t in times[1:])}), "
                      f"{threads} threads on the {cores} CPUs the job owns "
                      f"(cgroup cpu.max; {os.cpu_count()} hardware threads visible)"}


def family_table(fams, steps, step_s, bf16_mfma=("conv_bf16", "wgrad_bf16")):
    """Per-family roofline rows from the kernel timeline (sums over `steps` steps).  bf16_mfma: the families whose
    matrix work runs on the bf16 MFMA (priced against 2.5 PFLOP/s instead of the fp32-MFMA 157 TFLOP/s)."""
    rows = {}
    for name, f in fams.items():
        sec = f["ms"] / 1e3
        if sec <= 0:
            continue
        row = {"bound": f["bound"], "ms_per_step": f["ms"] / steps, "launches_per_step": f["launches"] / steps,
               "avg_launch_ms": f["ms"] / f["launches"], "step_time_share": sec / (step_s * steps)}
        if f["bound"] == "mfma":
            row["achieved"] = f["mfma_flops"] / sec / 1e12
            row["peak"], row["unit"] = (PEAK_BF16_MFMA_TFLOPS if name in bf16_mfma else PEAK_FP32_MFMA_TFLOPS), "TFLOP/s"
            row["algorithmic_speedup"] = f["alg_flops"] / f["mfma_flops"] if f["mfma_flops"] > 0 else 1.0
            row["hbm_gbs_algorithmic"] = f["hbm_bytes"] / sec / 1e9
        else:
            row["achieved"] = f["hbm_bytes"] / sec / 1e9
            row["peak"], row["unit"] = PEAK_HBM_GBS, "GB/s"
        row["frac"] = row["achieved"] / row["peak"]
        row["algorithmic_bytes_per_launch"] = f["hbm_bytes"] / f["launches"]
        rows[name] = row
    return rows


def self_launch(n: int) -> int:
    """Run this same command line under `python -m torch.distributed.run --nproc-per-node n` as a child
    process (one rank per GPU, rendezvous on 127.0.0.1 at a free port), relay its stdout / stderr, return its
    exit status.  The parent never initialises the GPU and never exec()s."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL needs it on this stack
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
    proc = subprocess.Popen(cmd, env=env)
    try:
        return proc.wait()
    except BaseException:
        proc.terminate()
        try:
            proc.wait(30)
        except subprocess.TimeoutExpired:
            proc.kill()
        raise


def main():
    if len(sys.argv) >= 7 and sys.argv[1] == "--cpu-baseline-child":      # (internal: see cpu_baseline)
        print(json.dumps(_cpu_baseline_impl(sys.argv[2], tuple(int(v) for v in sys.argv[3:6]), int(sys.argv[6]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--dtype", choices=("f32", "bf16"), default=None,
                    help="storage type of the activations: f32 = the reference's default arithmetic; bf16 = its "
                         "`--precision bf16` (bf16 activations, fp32 accumulation / statistics / parameters).  Default: "
                         "what BASELINE.json names for the config (bf16 for configs 2 and 4, f32 otherwise)")
    ap.add_argument("--detail", type=str, default="", help="write a per-convolution-call timing table to this file")
    ap.add_argument("--timeline", choices=("after", "in", "off"), default="after",
                    help="kernel timeline (roofline table): 'after' = a second pass of the same K steps right after "
                         "the timed region (default: the ~1,400 hipEventRecords per step cost 4-5 %% of a step, "
                         "so `value` is timed without them); 'in' = inside the timed region; 'off' = none")
    ap.add_argument("--predict", action="store_true",
                    help="time the predict path instead (models.py:430-450 + processor.py:111-129: eval forward, dRAM "
                         "up-projection x ess mask, percentages, resample + paste to an original grid) -- an extra "
                         "number for DESIGN.md, not the BASELINE metric")
    ap.add_argument("--graph", dest="graph", action="store_true", default=None,
                    help="capture the train step (fwd + loss + bwd + Adam) in one hipGraph and time its replays "
                         "(bit-identical to eager steps; the number then does not depend on how fast the host issues "
                         "launches): the default for bf16 storage and config 0 on one GPU and for bf16 data-parallel runs; fp32 "
                         "storage (one GPU, or data parallel over RCCL) picks it when a probe finds the host slow (config.graph_choice); a gloo rehearsal and "
                         "--detail / --predict default to eager steps")
    ap.add_argument("--no-graph", dest="graph", action="store_false", help="time eager steps on one GPU too")
    ap.add_argument("--graph-streams", type=int, choices=(1, 2), default=1,
                    help="branches of the captured step: 1 = single-stream capture (default); 2 keeps the weight-gradient side "
                         "stream inside the graph (measured equal on config 1, slower on config 0 / 2: DESIGN.md section 6)")
    ap.add_argument("--backend", choices=("nccl", "gloo"), default="nccl",
                    help="process-group backend; gloo + --share-gpu rehearses the N>1 control flow on a one-GPU box")
    ap.add_argument("--share-gpu", action="store_true", help="all ranks use cuda:0 (rehearsal only, not a measurement)")
    ap.add_argument("--force-dist", action="store_true",
                    help="keep the data-parallel collectives (RCCL) in place at world size 1")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # bare `python bench.py --gpus N` (the reference's `--ngpus N` alone starts N ranks, train.py:24,100-104):
        # start the N ranks as CHILD processes -- nothing in this parent has touched the GPU yet, and it never
        # will: it relays the children's output and exits with their status.
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # ONE JSON line on stdout, nothing else: RCCL prints a five-line version banner on rank 0's stdout when its
    # first communicator comes up, and nothing stops another native library from doing the like.  From here on file
    # descriptor 1 is stderr for everything in this process; the result goes to the saved descriptor at the end.
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with "
                         f"`python -m torch.distributed.run --nnodes=1 --nproc-per-node {args.gpus} bench.py --gpus {args.gpus} ...`")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    factory, B, dims, gflop_per_vol, act_elems, nparams = CONFIGS[args.config]
    if args.dtype is None:
        args.dtype = "bf16" if args.config in BF16_CONFIGS else "f32"

    cpu = None
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(factory, dims, B)          # first: the GPU burst below is then the tail of the run

    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    use_dist = world > 1 or args.force_dist
    if args.graph is None:
        # default on ONE GPU.  bf16 storage and the small config 0: the step replayed as a hipGraph (gap-bound steps: config
        # 2 104 -> 110 volumes/s, config 0 67-93 -> 99 in round 3).  fp32 configs 1 / 3 / 5: "auto" -- the eager two-stream
        # step is 0.2-0.6 ms faster than the replay of its capture (weight gradients overlap the data-gradient chain; a
        # second graph branch does not reproduce that: DESIGN.md section 6) AS LONG AS the host keeps ahead of the GPU; a cold
        # host (first process on a fresh box, seen once in round 5) issued the eager step in 23.7 instead of 7-10 ms and left
        # gaps: 42.3 ms for a step whose hipEvent median said 37.0.  So a probe of a few untimed eager steps measures the
        # host's issue time against the step time and picks the replay when the host needs more than 0.4 of the step
        # (`config.graph_choice` in the line says what was measured and picked; --graph / --no-graph override).
        # Data parallel: bf16 storage replays a graph WITH its collectives (RCCL through its C API is plain stream work,
        # distributed.DistContext.capturable); fp32 data-parallel steps over RCCL take the same probe; a gloo rehearsal runs eager.
        if args.predict or args.detail or (use_dist and args.backend != "nccl"):
            args.graph = False
        elif use_dist:
            # (fp32 data parallel over RCCL: the same probe as on one GPU, the choice agreed over the ranks -- a host that
            # cannot keep ahead of the GPU left the eager data-parallel step at 42.3 ms of wall clock against a hipEvent
            # median of 37.1, `profiles/r05_bench_config1_f32_forcedist_slow_host.json.log`)
            args.graph = True if (args.dtype == "bf16" or args.config == 0) else "auto"
        elif args.dtype == "bf16" or args.config == 0:
            args.graph = True
        else:
            args.graph = "auto"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)     # RCCL
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import bodyct_dram_emph_subtype_amd as dram
    from bodyct_dram_emph_subtype_amd import med3d, ops
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    dram.load_library()

    torch.manual_seed(0)
    kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
    module = getattr(med3d, factory)(**kw).to(device).train()
    if args.dtype == "bf16":
        module.storage_dtype = torch.bfloat16
    if args.config == 4:                     # BASELINE configs[4]: "... with activation checkpointing"
        module.activation_recompute = True
    dctx = None
    if use_dist:
        from bodyct_dram_emph_subtype_amd import distributed as ddist
        dctx = ddist.attach(module, force=args.force_dist)
    opt = FusedAdam(module.parameters(), lr=args.lr, capturable=bool(args.graph))
    batch = synth_batch(B, dims, rank, device)
    step = make_step(factory, module, opt, batch)
    graph_choice = None
    if args.graph == "auto":
        for _ in range(3):                   # plans, workspaces, optimizer state, allocator pool
            step()
        torch.cuda.synchronize()
        nprobe = 4
        w0 = module._engine.throttle_wait_s
        t0 = time.perf_counter()
        for _ in range(nprobe):
            step()
        issue_s = time.perf_counter() - t0 - (module._engine.throttle_wait_s - w0)
        torch.cuda.synchronize()
        total_s = time.perf_counter() - t0
        args.graph = issue_s > 0.4 * total_s
        if use_dist:                         # one form on every rank: the replay if any rank's host is slow
            flag = torch.tensor([1 if args.graph else 0], device=device, dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            args.graph = bool(int(flag.item()))
        graph_choice = (f"auto: the host issued {nprobe} untimed eager steps in {issue_s / nprobe * 1e3:.1f} ms/step of "
                        f"{total_s / nprobe * 1e3:.1f} ms/step -> " + ("hipGraph replay" if args.graph else "eager two-stream step"))
    if args.predict:
        from bodyct_dram_emph_subtype_amd import models as dmodels, processor
        if not factory.endswith("reg"):
            raise SystemExit("--predict needs a dRAM (reg) config: 2, 3 or 4")
        lm = dmodels.ScanRegLightningModule.__new__(dmodels.ScanRegLightningModule)
        torch.nn.Module.__init__(lm)
        lm.model = module.eval()
        image, lung, em, _, _ = batch
        D, H, W = dims
        pb = {"image": image[:, 0], "lung_mask": lung[:, 0] > 0, "ess_mask": (image[:, 0] < -0.5) & (lung[:, 0] > 0),
              "crop_slice": torch.tensor([[[3, 3 + D + 8], [5, 5 + H + 16], [7, 7 + W + 16]]] * B),
              "original_size": torch.tensor([[D + 20, H + 40, W + 40]] * B), "uid": [f"scan{i}" for i in range(B)]}

        def step():
            out = lm.predict_step(pb, 0)
            res = processor.build_outputs([out], want_u8=True)
            return res[0]["full_cle"].float().mean()
        args.timeline = "off"
    if args.graph and dctx is not None and not dctx.capturable:
        print("[bench] this process group's collectives run on the host (gloo): timing eager steps", file=sys.stderr)
        args.graph = False
    if args.graph:
        from bodyct_dram_emph_subtype_amd.graph import GraphedTrainStep
        from bodyct_dram_emph_subtype_amd.models import cls_train_loss, reg_train_loss
        if factory.endswith("cls"):
            cw = torch.full((6,), 1.0 / 6, device=device)
            pw = torch.full((3,), 1.0 / 3, device=device)

            def loss_fn(image, lung, em, cle, pse):
                return cls_train_loss(module(image, lung)[1], cle, pse, cw, pw)[0]
        else:
            cwt = torch.full((B,), 1.0 / 6, device=device)
            pwt = torch.full((B,), 1.0 / 3, device=device)

            def loss_fn(image, lung, em, cle, pse):
                dense, outs = module(image, lung)
                return reg_train_loss(dense, outs, lung, em, cle, pse, cwt, pwt)[0]
        eager_step = step
        try:
            graphed = GraphedTrainStep(module, opt, loss_fn, batch, warmup=2, streams=args.graph_streams)
            if graphed.graph is None:           # (data parallel: the runtime refused to record the collectives)
                args.graph = False
            else:
                step = lambda: graphed(*batch)      # noqa: E731
        except Exception as e:                  # capture refused (driver / allocator state): time eager steps, say so
            print(f"[bench] hipGraph capture failed ({type(e).__name__}: {e}); timing eager steps", file=sys.stderr)
            args.graph = False

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    timeline = ops.KernelTimeline(max_records=max(4096, 2048 * args.steps)) if (args.timeline != "off" and rank == 0) else None
    prof = None
    if args.detail:
        # per-call table: ONE stream (with the weight-gradient kernels overlapping the data-gradient chain on the
        # second stream a call's event interval would contain the time it shared the CUs with the other stream's
        # kernel: round 3's table summed conv_wgrad_w2d to 8.8 ms against 4.1 ms in the single-stream timeline)
        single_stream(True)
        step()
        barrier()
    if args.detail and rank == 0:
        prof = ops.KernelProfiler()
        ops.set_profiler(prof)
    if timeline and args.timeline == "in":
        timeline.start()
    # per-step hipEvent marks on the stream the step is launched on (median of the per-step
    # intervals); `value` stays the wall clock over exactly K steps between two barriers (driver contract)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    wait0 = module._engine.throttle_wait_s
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        loss = step()
        marks[i + 1].record()
    # the host has ISSUED the K steps (the GPU is still running them); minus the time the engine made it wait so as
    # not to run more than one step ahead (engine._throttle) = the time the host was busy issuing
    host_issue_s = time.perf_counter() - t0 - (module._engine.throttle_wait_s - wait0)
    barrier()
    dt = time.perf_counter() - t0
    step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(args.steps))
    median_step_ms = step_ms[len(step_ms) // 2] if len(step_ms) % 2 else 0.5 * (step_ms[len(step_ms) // 2 - 1] + step_ms[len(step_ms) // 2])
    ops.set_profiler(None)
    if args.detail:
        single_stream(False)
    tl_step_s = dt / args.steps
    # Data parallel: what do the collectives COST a step?  The same K steps once more WITHOUT them (context detached,
    # every rank on its own; same eager / graph form) right after the timed region: exposed_collective_ms = the
    # difference of the two step times, which by construction reconciles with the wall clock (the event-bracketed
    # sum of round 4 saw neither the cross-stream hand-offs nor RCCL's own launch overhead).
    plain_step_ms = None
    if dctx is not None and not args.predict:
        module._dist = None
        try:
            pstep = eager_step if args.graph else step
            if args.graph:
                g2 = GraphedTrainStep(module, opt, loss_fn, batch, warmup=2, streams=args.graph_streams)
                pstep = lambda: g2(*batch)      # noqa: E731
            for _ in range(max(2, args.warmup)):
                pstep()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(args.steps):
                pstep()
            torch.cuda.synchronize()
            plain_step_ms = (time.perf_counter() - tp) / args.steps * 1e3
        finally:
            module._dist = dctx
    if args.graph:
        step = eager_step                   # the timeline pass brackets individual launches: eager
    # every rank runs the second pass (its steps contain collectives); rank 0 records the timeline
    if args.timeline == "after" or (dctx is not None and args.timeline == "off"):
        # (data parallel with --timeline off: the second pass still runs, without the kernel timeline, so that EVERY N > 1
        # line carries exposed_collective_ms)
        # same K steps again, every kernel launch bracketed by hipEvents on its launch stream.  The timeline pass
        # runs the step on ONE stream (DRAM_WGRAD_STREAM=0): with the weight-gradient kernels overlapping the
        # data-gradient chain on a second stream a kernel's event interval would also contain the time it
        # shared its CUs with the other stream's kernel, and per-family fractions would mean nothing.
        single_stream(True)
        stats0 = dict(dctx.stats) if dctx is not None else None
        step()
        barrier()
        if dctx is not None:
            dctx.timing = True             # event pairs around every SyncBN statistic exchange (caller's stream)
        if timeline:
            timeline.start()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            step()
        barrier()
        tl_step_s = (time.perf_counter() - t1) / args.steps
        single_stream(False)
        if dctx is not None:
            exposed_ms = dctx.exposed_ms() / args.steps
            dctx.timing = False
    if timeline:
        timeline.stop()
    if world > 1:
        t = torch.tensor([dt], device=device if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    if rank == 0 and prof is not None:
        rows = sorted(prof.by_launch().items(), key=lambda kv: -kv[1]["ms"])
        with open(args.detail, "w") as f:
            f.write(f"# bench.py --config {args.config} --dtype {args.dtype} --detail: per-call hipEvent intervals, SINGLE-STREAM "
                    f"pass (DRAM_WGRAD_STREAM=0), {args.steps} steps; TF = direct-convolution FLOPs / time\n")
            for (fam, det), v in rows:
                tf = (v["flops"] / 1e12) / (v["ms"] / 1e3) if v["ms"] > 0 else 0.0
                f.write(f"{v['ms'] / args.steps:9.3f} ms/step  {v['launches'] // args.steps:3d}x  {tf:7.1f} TF  {fam}  {det}\n")
    if rank == 0:
        vols = args.steps * B * world
        step_s = dt / args.steps
        out = {
            "metric": f"CT volumes/sec ({'predict step + post-processing' if args.predict else 'train step'}, 1x{dims[0]}x{dims[1]}x{dims[2]})",
            "value": vols / dt,
            "unit": "volumes/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * step_s,
            "ms_per_step_median_hipevent": median_step_ms,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{'BASELINE configs[%d]' % args.config if args.config <= 4 else 'reference default job (train.py:21,30,42)'}: {factory} train step (fwd+loss+bwd+Adam), "
                                   f"batch {B}/GPU, 1x{dims[0]}x{dims[1]}x{dims[2]}, "
                                   f"{'fp32' if args.dtype == 'f32' else 'bf16 storage / fp32 accumulation, statistics and parameters'}, "
                                   f"inputs resident in HBM",
                       "global_batch": B * world, "parallelism": f"dp{world}", "hip_graph": bool(args.graph), "graph_choice": graph_choice,
                       "streams": (args.graph_streams if args.graph else
                                   (1 if (args.detail or ops.tuning_env("DRAM_WGRAD_STREAM", "1") == "0") else 2)),
                       "train_gflop_per_volume": gflop_per_vol},
            # host time to issue a step (waits for the GPU excluded); close to ms_per_step = the step is launch-bound
            "host_issue_ms_per_step": host_issue_s / args.steps * 1e3,
            "loss": float(loss.detach()),
            "peak_hbm_gb": torch.cuda.max_memory_allocated(device) / 1e9,
        }
        if dctx is not None:
            if args.timeline in ("after", "off"):    # counted over the eager second pass (a replayed graph issues none from the host)
                out["collectives_per_step"] = {k: (v - stats0[k]) / (args.steps + 1) for k, v in dctx.stats.items()}
            else:
                out["collectives_per_step"] = {k: v / (args.warmup + args.steps) for k, v in dctx.stats.items()}
            out["collective_transport"] = "rccl-c-api" if dctx._stat is not None else f"torch.distributed/{args.backend}"
            if plain_step_ms is not None:
                # step time with the collectives minus step time without them (same process, same form, measured
                # right after the timed region; on rank 0 -- dt is the max over ranks)
                out["plain_ms_per_step"] = plain_step_ms
                out["exposed_collective_ms"] = 1e3 * step_s - plain_step_ms
            if args.timeline in ("after", "off"):
                # event pairs around every SyncBN statistic exchange on the data path's stream, per step (single-stream
                # pass; the gradient buckets are asynchronous and not part of it): a LOWER bound of the above
                out["stat_exchange_ms_bracketed"] = exposed_ms
        if timeline:
            fams = timeline.families()
            bf16_fams = ("conv_bf16", "wgrad_bf16") + (("stem",) if (args.dtype == "bf16" and ops.tuning_env("DRAM_STEM_BF16", "1") != "0") else ())
            rows = family_table(fams, args.steps, tl_step_s, bf16_fams)
            default_dtype = "bf16" if args.config in BF16_CONFIGS else "f32"
            traffic, tsrc = measured_traffic(args.config) if (args.config in (1, 2) and args.dtype == default_dtype) else (None, None)
            for name, row in rows.items():
                key = TRAFFIC_KEY.get(name)
                row["traffic"] = float(traffic[key]["total"]) if (traffic and key in traffic) else None
            head = max(rows, key=lambda k: rows[k]["ms_per_step"])
            h = rows[head]
            kernel_ms = sum(r["ms_per_step"] for r in rows.values())
            mfma = sum(f["mfma_flops"] for f in fams.values())
            hbm = sum(f["hbm_bytes"] for f in fams.values())
            whole = {
                "timeline": args.timeline + (" (single-stream pass)" if args.timeline == "after" else ""),
                "ms_per_step_with_timeline": 1e3 * tl_step_s,
                "kernel_ms_per_step": kernel_ms, "timeline_coverage_of_step": kernel_ms / (1e3 * tl_step_s),
                # step-level fractions are priced against the UNinstrumented step time
                "executed_mfma_tflops": mfma / dt / 1e12,
                "executed_mfma_frac": sum(f["mfma_flops"] / (PEAK_BF16_MFMA_TFLOPS if n in bf16_fams else PEAK_FP32_MFMA_TFLOPS)
                                          for n, f in fams.items()) / dt / 1e12,
                "algorithmic_tflops": gflop_per_vol * vols / world / dt / 1e3,
                "hbm_gbs_algorithmic": hbm / dt / 1e9, "hbm_frac_algorithmic": hbm / dt / 1e9 / PEAK_HBM_GBS,
                # SURVEY.md §8d whole-step figure: fused-minimum activation traffic + 28 B/param
                "hbm_gbs_fused_minimum": (act_elems * (4 if args.dtype == "f32" else 2) * B + 28 * nparams) / step_s / 1e9,
                "hbm_gbs_measured": (traffic["_step_total_bytes"] / step_s / 1e9) if traffic and "_step_total_bytes" in traffic else None,
                "timeline_records_dropped": timeline.dropped,
            }
            if whole["hbm_gbs_measured"] is not None:
                whole["hbm_frac_measured"] = whole["hbm_gbs_measured"] / PEAK_HBM_GBS
            out["roofline"] = {
                "kernel": FAMILY_DESC.get(head, head), "family": head,
                "bound": h["bound"], "achieved": h["achieved"], "peak": h["peak"], "unit": h["unit"],
                "frac": h["frac"], "traffic": h["traffic"],
                "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
                "traffic_source": tsrc,
                "algorithmic_bytes_per_launch": h["algorithmic_bytes_per_launch"],
                "algorithmic_speedup": h.get("algorithmic_speedup"),
                "launches_per_step": h["launches_per_step"], "avg_launch_ms": h["avg_launch_ms"],
                "step_time_share": h["step_time_share"],
                "note": "achieved = EXECUTED MFMA FLOPs / time (mfma-bound) or ALGORITHMIC HBM bytes / time (hbm-bound) "
                        "from hipEvent pairs around every kernel launch, recorded on the launch stream over K steps "
                        "(whole_step.timeline says whether inside the timed region or in a second pass right after it); "
                        "algorithmic_speedup = "
                        "direct-convolution FLOPs / executed FLOPs (Winograd), never part of frac",
                "families": rows,
                "whole_step": whole,
            }
        if cpu is not None:
            out["cpu_baseline"] = cpu
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
