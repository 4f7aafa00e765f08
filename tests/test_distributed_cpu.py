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
        ctx = DistContext(bucket_bytes=40000)
        # gloo: torch.distributed's host-side collectives -- no own RCCL communicator, not capturable into a hipGraph
        assert ctx._stat is None and ctx._grad is None and ctx.capturable is False
        # the control-plane step of the RCCL transport: rank 0's 128-byte communicator id (zero bytes included) reaches
        # every rank intact (rccl.broadcast_id; the id itself comes from ncclGetUniqueId on a GPU box)
        from bodyct_dram_emph_subtype_amd import rccl
        raw = bytes([(3 * i) % 7 for i in range(128)])
        assert raw.count(0) > 10
        assert rccl.broadcast_id(raw if rank == 0 else None) == raw
        # --- SyncBN statistics: global mean/var from per-rank [sum, sum^2, count]; ranks hold DIFFERENT counts
        rows = 5 + 2 * rank
        x = torch.randn(rows, 8, dtype=torch.float64)            # local "activations" [rows, C]
        flat = torch.cat([x.sum(0), (x * x).sum(0), torch.tensor([float(rows)], dtype=torch.float64)])
        ctx.all_reduce_stats(flat)                                # in place
        gs, cnt = flat[:16].view(2, 8), float(flat[16])
        allx = [torch.zeros(5 + 2 * r, 8, dtype=torch.float64) for r in range(world)]
        dist.all_gather_object(allx, x)
        full = torch.cat(allx)
        ok = cnt == float(full.shape[0]) and torch.allclose(gs[0] / cnt, full.mean(0)) and \
            torch.allclose(gs[1] / cnt - (gs[0] / cnt) ** 2, full.var(0, unbiased=False))
        w = ctx.all_reduce_stats_async(flat[:16])                 # the backward form: asynchronous, then wait()
        w.wait()
        ok = ok and torch.allclose(flat[0:8] / world, full.sum(0))
        # --- gradient arena: large parameters are written in place (views in parameter order), finished
        #     ranges are all-reduced asynchronously from the end; small ones travel in one extra bucket
        params = [(f"p{i}", torch.empty(5000 + 64 * i) if i != 3 else torch.empty(17)) for i in range(7)]
        ctx.bind(params)
        n0 = dict(ctx.stats)
        ctx.begin_backward("cpu")
        grads, local = {}, {}
        for name, p in params:
            out = ctx.grad_out(name)
            assert (out is None) == (p.numel() < 4096)
            g = torch.randn(p.shape)
            local[name] = g.clone()
            if out is not None and name != "p2":
                out.copy_(g)                                      # "the wgrad kernel wrote the arena view"
                grads[name] = out
            else:
                grads[name] = g                                   # produced elsewhere: copied in by grads_ready
        ctx.grads_ready(grads, ["p6", "p5"])                      # 2 x 20 KB >= bucket: launched here
        launched_early = ctx.stats["grad_allreduce"] - n0["grad_allreduce"]
        ctx.grads_ready(grads, ["p4", "p3", "p1"])                # p2 missing: range [p4] only is contiguous-final
        ctx.grads_ready(grads, ["p2", "p0", "p0"])                # duplicate names are ignored
        ctx.finish(grads)
        ok = ok and launched_early == 1 and ctx.stats["grad_allreduce"] - n0["grad_allreduce"] <= 4
        lo, nbytes = ctx.last_arena
        for k in local:
            g_all = [torch.zeros_like(local[k]) for _ in range(world)]
            dist.all_gather(g_all, local[k])
            ok = ok and torch.allclose(grads[k], sum(g_all) / world, atol=1e-6)
            if local[k].numel() >= 4096:
                ok = ok and lo <= grads[k].data_ptr() < lo + nbytes    # the averaged gradient IS the arena view
        # an interrupted step must not leak state into the next one
        ctx.begin_backward("cpu")
        ctx.grads_ready({"p6": ctx.grad_out("p6")}, ["p6"])
        ctx.begin_backward("cpu")
        ok = ok and not ctx._done and not ctx._inflight and ctx._hi == len(ctx._order)
        ctx.finish({})
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


def _strategy_worker(rank, world, port, q):
    """Lightning seam (B4): a stand-in DDPStrategy with the 1.9 hook names; DramDDPStrategy must attach the engine's
    DistContext to module.model (broadcasting rank 0's parameters) and return the module UNWRAPPED."""
    import types
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        class DDPStrategy:                                   # pytorch_lightning.strategies.DDPStrategy (1.9) outline
            def __init__(self, **kw):
                self._ddp_kwargs = kw
                self.model = None

            def _setup_model(self, model):
                raise AssertionError("the stock strategy would wrap the module in DistributedDataParallel")

            def _register_ddp_hooks(self):
                raise AssertionError("no DDP hooks for the fused engine")

            def configure_ddp(self):
                self.model = self._setup_model(types.SimpleNamespace(module=self.lightning_module))
                self._register_ddp_hooks()

        pl = types.ModuleType("pytorch_lightning")
        st = types.ModuleType("pytorch_lightning.strategies")
        st.DDPStrategy = DDPStrategy
        pl.strategies = st
        sys.modules["pytorch_lightning"], sys.modules["pytorch_lightning.strategies"] = pl, st
        from bodyct_dram_emph_subtype_amd import med3d
        from bodyct_dram_emph_subtype_amd.ddp_strategy import make_ddp_strategy
        torch.manual_seed(50 + rank)
        lm = torch.nn.Module()
        lm.model = med3d.resnet18segreg()
        s = make_ddp_strategy(process_group_backend="gloo")
        s.lightning_module = lm
        s.configure_ddp()
        ok = s._ddp_kwargs == dict(process_group_backend="gloo", find_unused_parameters=False)
        ok = ok and s.model.module is lm and lm.model._dist is s.dram_context and s.dram_context.world == world
        w = lm.model.conv1.weight.data
        ws = [torch.zeros_like(w) for _ in range(world)]
        dist.all_gather(ws, w)
        ok = ok and torch.equal(ws[0], ws[1])                 # attach() broadcast rank 0's parameters
        bad = torch.nn.Module()
        s.lightning_module = bad
        try:
            s.configure_ddp()
            ok = False
        except RuntimeError:
            pass
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


def test_lightning_ddp_strategy_binding_world2_gloo():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 27500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_strategy_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(60)
    assert sorted(res) == [(0, True), (1, True)]


def test_bench_gpus_n_bare_command_starts_child_ranks():
    """bench.py --gpus 2 without WORLD_SIZE must start its own ranks (no GPU here: each child stops at the
    "needs an MI355X" check -- which proves rank processes were started under torch.distributed.run -- and
    the parent exits with their non-zero status instead of complaining about WORLD_SIZE)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if torch.cuda.is_available():
        pytest.skip("covered by the -m gpu rehearsal on a GPU box")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--backend", "gloo", "--share-gpu"], env=env, capture_output=True, text=True, timeout=600, cwd="/tmp")
    assert r.returncode != 0
    assert "WORLD_SIZE=" not in r.stderr, r.stderr[-2000:]
    # (torchrun tears the sibling down as soon as one rank exits, so the message may appear once or twice)
    assert r.stderr.count("bench.py needs an MI355X") >= 1 and "torch.distributed.elastic" in r.stderr, r.stderr[-2000:]
