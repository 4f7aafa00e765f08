"""CPU, world_size 2 over gloo: the host logic of the data-parallel path
(bodyct-dram-emph-subtype_amd/distributed.py) -- SyncBN statistic exchange, bucketed
asynchronous gradient averaging, parameter broadcast -- on CPU tensors, and the oracle's
N-rank DDP+SyncBN emulation against a literal 2-process run."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bodyct_dram_emph_subtype_amd.distributed import DistContext, broadcast_parameters
        torch.manual_seed(100 + rank)
        ctx = DistContext(bucket_bytes=4096)
        # --- SyncBN statistics: global mean/var from per-rank [sum, sum^2]
        x = torch.randn(5, 8, dtype=torch.float64)            # local "activations" [rows, C]
        sums = torch.stack([x.sum(0), (x * x).sum(0)])
        gs, cnt = ctx.sync_bn_stats(sums, 5.0)
        allx = [torch.zeros_like(x) for _ in range(world)]
        dist.all_gather(allx, x)
        full = torch.cat(allx)
        ok = cnt == 10.0 and torch.allclose(gs[0] / cnt, full.mean(0)) and \
            torch.allclose(gs[1] / cnt - (gs[0] / cnt) ** 2, full.var(0, unbiased=False))
        ok = ok and torch.equal(sums, torch.stack([x.sum(0), (x * x).sum(0)]))   # input untouched
        # --- bucketed async gradient mean (several buckets, names arriving in groups)
        grads = {f"p{i}": torch.randn(300 + 17 * i) for i in range(7)}
        local = {k: v.clone() for k, v in grads.items()}
        ctx.grads_ready(grads, ["p6", "p5"])
        ctx.grads_ready(grads, ["p4", "p3", "p2"])
        ctx.grads_ready(grads, ["p1", "p0", "p0"])           # duplicate names are ignored
        ctx.finish(grads)
        for k in local:
            g_all = [torch.zeros_like(local[k]) for _ in range(world)]
            dist.all_gather(g_all, local[k])
            ok = ok and torch.allclose(grads[k], sum(g_all) / world, atol=1e-6)
        # --- parameter broadcast
        lin = torch.nn.Sequential(torch.nn.Linear(4, 3), torch.nn.BatchNorm1d(3))
        broadcast_parameters(lin)
        w_all = [torch.zeros_like(lin[0].weight.data) for _ in range(world)]
        dist.all_gather(w_all, lin[0].weight.data)
        ok = ok and torch.equal(w_all[0], w_all[1])
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_dist_context_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def _ddp_ref_worker(rank, world, port, q):
    """literal 2-process DDP-style run of the ORACLE network on CPU: per-rank forward with
    all-reduced BN statistics emulated by running BN over the gathered batch is not possible
    with torch's CPU ops, so this worker checks the *loss/gradient averaging* identity the
    emulation relies on: grads of mean_r(loss_r) == mean_r(grads of loss_r) for a model
    without batch coupling."""
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        lin = torch.nn.Linear(6, 2)
        g = torch.Generator().manual_seed(5)
        xs = [torch.randn(3, 6, generator=g) for _ in range(world)]
        loss = lin(xs[rank]).square().mean()
        loss.backward()
        gw = lin.weight.grad.clone()
        dist.all_reduce(gw)
        gw /= world
        lin.zero_grad()
        (sum(lin(x).square().mean() for x in xs) / world).backward()
        q.put((rank, bool(torch.allclose(gw, lin.weight.grad, atol=1e-6))))
    finally:
        dist.destroy_process_group()


def test_ddp_mean_identity_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_ddp_ref_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_oracle_ddp_emulation_equals_single_process_when_loss_is_additive():
    """ddp_emulated_grads with N ranks of batch 1 == plain big-batch gradient for a loss that is
    a mean over samples (sanity of the fixture generator used by the multi-GPU parity test)."""
    from oracle import med3d_oracle as orc
    from bodyct_dram_emph_subtype_amd import med3d
    torch.manual_seed(3)
    m = med3d.resnet18segreg()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    g = torch.Generator().manual_seed(9)
    xs = [torch.randn(1, 1, 16, 16, 16, generator=g) for _ in range(2)]
    grads, total = orc.ddp_emulated_grads(sd, xs, [None, None], "resnet18segreg",
                                          lambda r, d, o: o[0].sum() + o[1].sum())
    names = [n for n, _ in m.named_parameters()]
    leaves = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    d, o = orc.forward(leaves, torch.cat(xs), None, "resnet18segreg", train=True)
    ((o[0].sum() + o[1].sum()) / 2).backward()
    for n in ("conv1.weight", "layer3.0.conv1.weight", "fcs.1.weight"):
        assert torch.allclose(grads[n], leaves[n].grad, rtol=1e-4, atol=1e-7), n
