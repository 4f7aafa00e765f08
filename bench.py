"""bench.py -- CT volumes/sec of the Med3D + dRAM train step on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--config 0|1|2|3|4]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = forward + loss + backward + (N>1: RCCL gradient all-reduce, SyncBN exchanges) +
fused Adam update on one batch of synthetic volumes already resident in HBM
(SURVEY.md §8d recipe).  Default workload = BASELINE.json configs[1]:
conf/med3d18.yaml (resnet18segcls), batch 2 per GPU, 1x128x256x256, fp32.

Prints ONE JSON line (rank 0) with the driver's contract plus
  roofline     -- dominant single kernel of the step (the one with the largest HIP-event time
                  among the conv kernels; on config 1 the fused in-plane Winograd conv),
                  ALGORITHMIC (direct-convolution) FLOPs / HIP-event time measured inside the
                  timed region; `executed_*` prices the MFMA products the kernel really issues
  cpu_baseline -- the CPU oracle (oracle/med3d_oracle.py, torch CPU ops) on a bounded
                  sample of the same workload, host cores stated  (N=1, rank 0 only)
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CONFIGS = {
    # id: (factory, per-GPU batch, (D,H,W), train GFLOP per volume [SURVEY.md §8a])
    0: ("resnet34segcls", 1, (64, 128, 128), 1259.3),
    1: ("resnet18segcls", 2, (128, 256, 256), 6943.0),
    2: ("resnet18segreg", 2, (128, 256, 256), 6941.6),
    3: ("resnet50segreg", 1, (128, 256, 256), 10313.4),
    # configs[4] geometry (full-resolution volume) in fp32 without activation checkpointing: a capacity /
    # int32-offset check of the kernels at 8x the voxels, not a BASELINE metric (that one asks for bf16)
    4: ("resnet50segreg", 1, (256, 512, 512), 8 * 10313.4),
}
PEAK_FP32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md, Chip-level parameters


# profiler span -> (kernel description, key in profiles/r01_pmc_hbm_traffic.json,
#                   algorithmic MACs / executed MFMA MACs)
ROOFLINE_KERNELS = {
    "conv_wino2d_kernel": ("conv_wino2d_kernel<NJ> (fused in-plane Winograd F(2x2,3x3) x direct-z conv, fwd + dgrad, "
                           "fp32 MFMA 32x32x2)", "conv_wino2d", 54.0 / 24.0),
    "conv_igemm_kernel": ("conv_igemm*_kernel family (direct implicit-GEMM conv fwd + dgrad, fp32 MFMA 32x32x2)",
                          "conv_igemm", 1.0),
}


def measured_traffic(key):
    """HBM bytes per launch of the roofline kernel from rocprofv3 PMC passes (FETCH_SIZE x2
    gfx950 correction + WRITE_SIZE, separate passes), recorded offline for this exact command
    (config 1) in profiles/r01_pmc_hbm_traffic.json (tools/profile_round.sh) -- PMC counters
    cannot be read from inside bench.py."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_hbm_traffic.json")
    try:
        with open(path) as f:
            return float(json.load(f)[key]["total"])
    except Exception:  # noqa: BLE001
        return None


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
    image, lung, em, cle, pse = batch
    if factory.endswith("cls"):
        cw = torch.full((6,), 1.0 / 6, device=image.device)
        pw = torch.full((3,), 1.0 / 3, device=image.device)

        def step():
            opt.zero_grad(set_to_none=True)
            _, outs = module(image, lung)
            # models.py:253-258 -- class-weighted CE on [B,6] and [B,3] (K16: torch glue)
            loss = F.cross_entropy(outs[0], cle, weight=cw) + F.cross_entropy(outs[1], pse, weight=pw)
            loss.backward()
            opt.step()
            return loss
        return step
    from bodyct_dram_emph_subtype_amd.models import reg_train_loss
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


def cpu_baseline(factory):
    """Oracle train step (fwd + loss + bwd + Adam) on the host cores, bounded sample."""
    from oracle import med3d_oracle as orc
    from bodyct_dram_emph_subtype_amd import med3d
    dims = (64, 256, 256)          # half of one 128x256x256 volume's voxels, batch 1
    frac = 0.5
    torch.manual_seed(0)
    kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
    m = getattr(med3d, factory)(**kw)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    names = [n for n, _ in m.named_parameters()]
    image, lung, em, cle, pse = synth_batch(1, dims, 0, "cpu")
    mom = {n: (torch.zeros_like(sd[n]), torch.zeros_like(sd[n])) for n in names}
    cores = torch.get_num_threads()
    t0 = time.perf_counter()
    leaves = {k: (v.requires_grad_(True) if k in names else v) for k, v in sd.items()}
    dense, outs = orc.forward(leaves, image, lung, factory, train=True)
    if factory.endswith("cls"):
        loss, _ = orc.cls_train_loss(outs, cle, pse, torch.full((6,), 1 / 6), torch.full((3,), 1 / 3))
    else:
        loss, _ = orc.reg_train_loss(dense, outs, lung, em, cle, pse, torch.full((1,), 1 / 6), torch.full((1,), 1 / 3))
    loss.backward()
    with torch.no_grad():
        for n in names:
            orc.adam_step(leaves[n], leaves[n].grad, mom[n][0], mom[n][1], 1, 1e-4)
    dt = time.perf_counter() - t0
    return {"value": frac / dt, "unit": "volumes/sec", "cores": cores, "kind": "port",
            "sample": f"1 train step of {factory} (oracle, torch CPU ops) on 1x1x{dims[0]}x{dims[1]}x{dims[2]} "
                      f"= {frac} volume of 128x256x256, {dt:.1f} s, scaled by voxel count"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--config", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--lr", type=float, default=1e-4)
    ap.add_argument("--detail", type=str, default="", help="write a per-launch-geometry timing table to this file")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    import bodyct_dram_emph_subtype_amd as dram
    from bodyct_dram_emph_subtype_amd import med3d, ops
    from bodyct_dram_emph_subtype_amd.optim import FusedAdam
    dram.load_library()

    factory, B, dims, gflop_per_vol = CONFIGS[args.config]
    torch.manual_seed(0)
    kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
    module = getattr(med3d, factory)(**kw).to(device).train()
    if world > 1:
        from bodyct_dram_emph_subtype_amd import distributed as ddist
        ddist.attach(module)
    opt = FusedAdam(module.parameters(), lr=args.lr)
    batch = synth_batch(B, dims, rank, device)
    step = make_step(factory, module, opt, batch)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    prof = ops.KernelProfiler()
    ops.set_profiler(prof)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    dt = time.perf_counter() - t0
    ops.set_profiler(None)
    if world > 1:
        t = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t)
    summ = prof.summary()
    if rank == 0 and args.detail:
        rows = sorted(prof.by_launch().items(), key=lambda kv: -kv[1]["ms"])
        with open(args.detail, "w") as f:
            for (fam, det), v in rows:
                tf = (v["flops"] / 1e12) / (v["ms"] / 1e3) if v["ms"] > 0 else 0.0
                f.write(f"{v['ms'] / args.steps:9.3f} ms/step  {v['launches'] // args.steps:3d}x  {tf:7.1f} TF  {fam}  {det}\n")
    if rank == 0:
        vols = args.steps * B * world
        fam = max(ROOFLINE_KERNELS, key=lambda k: summ.get(k, {}).get("ms", 0.0))
        kdesc, tkey, ratio = ROOFLINE_KERNELS[fam]
        s = summ.get(fam, dict(launches=0, ms=0.0, flops=0.0))
        achieved = (s["flops"] / 1e12) / (s["ms"] / 1e3) if s["ms"] > 0 else 0.0
        out = {
            "metric": "CT volumes/sec (train step, 1x128x256x256)",
            "value": vols / dt,
            "unit": "volumes/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"BASELINE configs[{args.config}]: {factory} train step (fwd+loss+bwd+Adam), "
                                   f"batch {B}/GPU, 1x{dims[0]}x{dims[1]}x{dims[2]}, fp32, inputs resident in HBM",
                       "global_batch": B * world, "parallelism": f"dp{world}",
                       "train_gflop_per_volume": gflop_per_vol},
            "loss": float(loss.detach()),
            "peak_hbm_gb": torch.cuda.max_memory_allocated(device) / 1e9,
            "roofline": {
                "kernel": kdesc,
                "bound": "mfma", "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / PEAK_FP32_MFMA_TFLOPS,
                "note": "achieved = algorithmic (direct-convolution) FLOPs / time; a Winograd kernel executes "
                        "fewer MFMA products than that, so frac may exceed 1 -- executed_* is the matrix-pipe view",
                "algorithmic_over_executed_flops": ratio,
                "executed_tflops": achieved / ratio, "executed_frac": achieved / ratio / PEAK_FP32_MFMA_TFLOPS,
                "traffic": measured_traffic(tkey) if args.config == 1 else None,
                "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, profiles/r01_pmc_hbm_traffic.json)",
                "launches_per_step": s["launches"] / max(args.steps, 1),
                "avg_launch_ms": s["ms"] / max(s["launches"], 1),
                "step_time_share": (s["ms"] / 1e3) / dt if dt > 0 else 0.0,
                "whole_step_tflops": gflop_per_vol * vols / world / dt / 1e3,
                "families": {k: {"ms_per_step": v["ms"] / args.steps,
                                 "tflops": (v["flops"] / 1e12) / (v["ms"] / 1e3) if v["ms"] > 0 else 0.0}
                             for k, v in summ.items()},
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(factory)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
