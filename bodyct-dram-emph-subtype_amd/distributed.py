"""Data parallelism for the Med3D engine: one process per GPU, RCCL over xGMI.

Stands in for what the reference gets from Lightning's ``DDPStrategy`` +
``sync_batchnorm=True`` (reference train.py:70,100-104):

  C1  gradient mean over ranks     -> the weight-gradient kernels write straight into ONE flat
                                      arena (a view per parameter, parameter order); contiguous
                                      ranges of it are all-reduced *asynchronously* from inside the
                                      engine's backward as soon as >= bucket_bytes of it are final,
                                      overlapped with the rest of backward.  No concatenation copy.
  C2  SyncBN forward statistics    -> one in-place all-reduce of [sum, sum^2, count] (2C+1 doubles);
                                      the global count stays on the device (ranks may hold different
                                      batch sizes, as under torch SyncBatchNorm)
  C3  SyncBN backward sums         -> one in-place all-reduce of [sum g, sum g*xhat] (2C doubles),
                                      launched asynchronously; the engine runs the PREVIOUS unit's
                                      weight-gradient kernel while it is in flight
  C4  per-step buffer broadcast    -> dropped: SyncBN keeps running stats identical
  C5  initial parameter broadcast  -> broadcast_parameters()

Collectives per train step: 2 x (number of BN layers) statistic all-reduces (22/38/54 layers for
ResNet-18/34/50: 44/76/108, each <= 32 KB, latency-bound; the backward half is hidden behind a
weight-gradient kernel) + ceil(parameter bytes / bucket_bytes) + 1 gradient all-reduces (R18: 5 + 1).

Transport.  Under the ``nccl`` backend (= RCCL on ROCm) the collectives do NOT go through ``torch.distributed``: the
context owns two RCCL communicators reached through librccl's C API (rccl.py).  The statistic exchanges are enqueued on
the DATA stream itself -- the stream the producing fold kernel was launched on -- so an exchange is one more kernel in
the data path's queue: no second stream, no event hand-off (ProcessGroupNCCL's two hand-offs per call were +10 / +15 /
+38 % on the config 1 / config 2 bf16 / ResNet-50 bf16 step at world size 1, round 4).  The gradient buckets go to a
communication stream forked from the producing stream and joined at the end of backward with plain events.  All of it is
ordinary stream work, so the data-parallel step can be captured into a hipGraph (``capturable``).  ``gloo`` groups (CPU
or device tensors: the tests) keep ``torch.distributed``'s collectives; DRAM_DIST_TRANSPORT=torch under DRAM_TUNING=1
forces that path under ``nccl`` too (A/B).  ``force=True`` (or DRAM_DIST_FORCE=1) keeps every collective in place at
world_size 1, which is how the RCCL code path is exercised on a one-GPU box.  The engine's fused
conv-BN units are invisible to ``SyncBatchNorm.convert_sync_batchnorm``, hence
this module.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch
import torch.distributed as dist

Tensor = torch.Tensor
SMALL_NUMEL = 4096      # parameters below this (BN affine, biases, 1x1x1 heads) travel in one extra small bucket


class _Done:
    def wait(self):
        return True


class _TimedWork:
    """work handle whose wait() is bracketed by two events on the caller's stream"""

    def __init__(self, work, ctx):
        self.work, self.ctx = work, ctx

    def wait(self):
        a = self.ctx._mark()
        r = self.work.wait()
        if a is not None:
            self.ctx._timed.append((a, self.ctx._mark()))
        return r


class DistContext:
    def __init__(self, process_group=None, sync_bn: bool = True, bucket_bytes: int = 32 << 20,
                 average: bool = True, force: Optional[bool] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        if force is None:
            force = os.environ.get("DRAM_DIST_FORCE", "0") == "1"
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.active = self.world > 1 or bool(force)
        self.sync_bn = sync_bn
        self.bucket_bytes = bucket_bytes
        self.average = average
        self._backend = dist.get_backend(process_group)
        # SyncBN statistics travel on their OWN communicator: a latency-critical 2C-double exchange the data path
        # waits for must not queue behind a >= 32 MB gradient bucket that itself waits for the side stream's
        # weight-gradient kernels.
        self.stat_pg = process_group
        self._stat = self._grad = self._comm_stream = None       # own RCCL communicators (rccl.py) + the bucket stream
        if self.active and self._backend == "nccl" and self._transport() == "rccl":
            from . import rccl
            # (every rank takes the same branch: a failure to come up is agreed upon below before anybody uses either path)
            try:
                self._stat = rccl.Communicator(process_group)    # used on the data stream itself
                self._grad = rccl.Communicator(process_group)    # used on _comm_stream
                self._comm_stream = torch.cuda.Stream()
                ok = 1
            except (RuntimeError, OSError) as exc:
                import sys
                print(f"distributed: own RCCL communicators did not come up on rank {self.rank} ({exc}); "
                      "falling back to torch.distributed's collectives", file=sys.stderr)
                ok = 0
            if self.world > 1:
                flag = torch.tensor([ok], device="cuda", dtype=torch.int32)
                dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=process_group)
                ok = int(flag.item())
            if not ok:
                for c in (self._stat, self._grad):
                    if c is not None:
                        c.destroy()
                self._stat = self._grad = self._comm_stream = None
        if self._stat is None and self.active and (self.world > 1 or self._backend == "nccl"):
            # torch.distributed transport (gloo; nccl under DRAM_DIST_TRANSPORT=torch): under RCCL the group gets a
            # high-priority HIP stream, torch hands the results over with events
            self.stat_pg = self._new_stat_group(process_group)
        self.timing = False                                              # bench.py --force-dist: time the exchanges
        self._timed: List[tuple] = []
        self._slots: Dict[str, Tuple[int, int, Tuple[int, ...]]] = {}    # big params: name -> (index, offset, shape)
        self._order: List[Tuple[str, int, int]] = []                     # (name, offset, numel) in parameter order
        self._total = 0
        self.stats = dict(bn_allreduce=0, grad_allreduce=0)              # collectives issued since construction
        self.last_arena = None                                           # (ptr, bytes) of the last finished step's arena
        self._reset()

    @staticmethod
    def _transport() -> str:
        from . import ops, rccl
        want = ops.tuning_env("DRAM_DIST_TRANSPORT", "rccl")
        return "rccl" if (want != "torch" and rccl.available()) else "torch"

    @property
    def capturable(self) -> bool:
        """Can a step with these collectives be captured into a hipGraph?  Yes when they are plain stream work (own
        RCCL communicators) or absent (inactive context)."""
        return (not self.active) or self._stat is not None

    def _new_stat_group(self, process_group):
        ranks = dist.get_process_group_ranks(process_group) if process_group is not None else None
        if self._backend == "nccl":
            try:
                opts = dist.ProcessGroupNCCL.Options(is_high_priority_stream=True)
                return dist.new_group(ranks=ranks, backend="nccl", pg_options=opts)
            except Exception:  # noqa: BLE001  (older torch: no options object) -- still a separate communicator
                return dist.new_group(ranks=ranks, backend="nccl")
        return dist.new_group(ranks=ranks, backend=self._backend)

    def _mark(self):
        if not self.timing or not torch.cuda.is_available():
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def exposed_ms(self, reset: bool = True) -> float:
        """Sum of the intervals the CALLER'S stream spent between issuing a statistic exchange and being allowed
        to continue (forward: the blocking call; backward: work.wait(); own communicators: the RCCL kernel's span on
        the data stream), over everything timed since the last reset.  Synchronises the device.  A lower bound of
        what the collectives cost a step -- bench.py reports the step-time difference to the plain step next to it."""
        torch.cuda.synchronize()
        total = sum(a.elapsed_time(b) for a, b in self._timed)
        if reset:
            self._timed = []
        return total

    # ---------------------------------------------------------------- layout
    def bind(self, named_params):
        """Lay the large parameters out in one flat gradient arena, in parameter order (backward
        produces them in roughly the reverse order, so finished ranges grow from the end)."""
        self._slots, self._order, off = {}, [], 0
        for name, p in named_params:
            if p.numel() >= SMALL_NUMEL:
                self._slots[name] = (len(self._order), off, tuple(p.shape))
                self._order.append((name, off, p.numel()))
                off += (p.numel() + 63) // 64 * 64        # 256-byte aligned views
        self._total = off

    def _reset(self):
        self._arena: Optional[Tensor] = None
        self._ready: List[bool] = []
        self._hi = 0                  # slots [_hi, end) are already being reduced
        self._small: List[str] = []
        self._inflight: List[tuple] = []
        self._done = set()

    def begin_backward(self, device):
        """Start of every engine.backward: forget whatever an interrupted step left behind (a caught
        OOM must not make the next step skip or wait on stale work) and take a FRESH arena -- the
        previous one may still back p.grad (gradient accumulation without zero_grad)."""
        self._reset()
        if self._total:
            self._arena = torch.empty((self._total,), device=device, dtype=torch.float32)
        self._ready = [False] * len(self._order)
        self._hi = len(self._order)

    def grad_out(self, name: str) -> Optional[Tensor]:
        """The arena view the weight-gradient kernel of `name` should write (None: not an arena parameter)."""
        s = self._slots.get(name)
        if s is None or self._arena is None:
            return None
        _, off, shape = s
        n = 1
        for d in shape:
            n *= d
        return self._arena[off:off + n].view(shape)

    # ---------------------------------------------------------------- collectives
    def _all_reduce(self, t: Tensor, avg: bool, async_op: bool):
        if avg and self._backend == "nccl":
            return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.pg, async_op=async_op), True
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=async_op), False

    def all_reduce_stats(self, flat: Tensor):
        """In-place SUM of a float64 statistics buffer over ranks (forward: the consumer needs it now)."""
        if not (self.sync_bn and self.active):
            return
        self.stats["bn_allreduce"] += 1
        a = self._mark()
        if self._stat is not None:
            self._stat.all_reduce(flat)                 # on the data stream: the consumer is simply the next kernel
        else:
            dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.stat_pg)
        if a is not None:
            self._timed.append((a, self._mark()))

    def all_reduce_stats_async(self, flat: Tensor):
        """Same, asynchronous: returns a handle whose wait() orders the current stream (RCCL) or the
        host (gloo) after the reduction.  The caller launches independent kernels in between."""
        if not (self.sync_bn and self.active):
            return _Done()
        self.stats["bn_allreduce"] += 1
        if self._stat is not None:
            # own communicator: enqueued on the data stream, in order -- nothing to wait for on the host; the GPU stays
            # busy with the weight-gradient kernels of the engine's second stream meanwhile
            a = self._mark()
            self._stat.all_reduce(flat)
            if a is not None:
                self._timed.append((a, self._mark()))
            return _Done()
        work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.stat_pg, async_op=True)
        return _TimedWork(work, self) if self.timing else work

    # ---------------------------------------------------------------- gradients
    def grads_ready(self, grads: Dict[str, Tensor], names: List[str]):
        """Called by the engine's backward when `names` have their final local gradients."""
        if not self.active:
            return
        for n in names:
            if n in self._done or n not in grads:
                continue
            self._done.add(n)
            s = self._slots.get(n)
            if s is None or self._arena is None:
                self._small.append(n)
                continue
            view = self.grad_out(n)
            if grads[n].data_ptr() != view.data_ptr():          # produced elsewhere: bring it into the arena
                view.copy_(grads[n])
                grads[n] = view
            self._ready[s[0]] = True
        lo = self._hi
        while lo > 0 and self._ready[lo - 1]:
            lo -= 1
        if lo < self._hi:
            a = self._order[lo][1]
            b = self._order[self._hi - 1][1] + self._order[self._hi - 1][2]
            if (b - a) * 4 >= self.bucket_bytes or lo == 0:
                self._launch_range(lo)

    def _launch_range(self, lo: int):
        a = self._order[lo][1]
        b = self._order[self._hi - 1][1] + self._order[self._hi - 1][2]
        seg = self._arena[a:b]
        work, scaled = self._bucket_all_reduce(seg)
        self.stats["grad_allreduce"] += 1
        self._inflight.append((work, seg, scaled))
        self._hi = lo

    def _bucket_all_reduce(self, seg: Tensor):
        """Mean (or sum) of a finished gradient range over ranks, asynchronous to the launching stream."""
        if self._grad is None:
            return self._all_reduce(seg, self.average, True)
        from . import rccl
        # fork: the communication stream continues from the point the producing stream (the weight-gradient stream, or
        # the caller's) has reached; joined in finish().  Plain events: capturable.
        ready = torch.cuda.Event()
        ready.record()
        self._comm_stream.wait_event(ready)
        seg.record_stream(self._comm_stream)
        self._grad.all_reduce(seg, rccl.AVG if self.average else rccl.SUM, stream=self._comm_stream.cuda_stream)
        return None, True

    def finish(self, grads: Dict[str, Tensor]):
        """Flush what is left, wait for every bucket; the small parameters go in one extra bucket."""
        if not self.active:
            return
        if self._hi > 0:
            # parameters that never reported (frozen ones) leave undefined arena bytes nobody reads
            self._launch_range(0)
        small = [n for n in self._small if n in grads]
        flat = None
        if small:
            flat = torch.cat([grads[n].reshape(-1) for n in small])
            work, scaled = self._bucket_all_reduce(flat)
            self.stats["grad_allreduce"] += 1
            self._inflight.append((work, flat, scaled))
        if self._grad is not None:
            torch.cuda.current_stream().wait_stream(self._comm_stream)        # join: every bucket is final from here on
        for work, seg, scaled in self._inflight:
            if work is not None:
                work.wait()
            if self.average and not scaled:
                seg.div_(self.world)
        if small:
            off = 0
            for n in small:
                k = grads[n].numel()
                grads[n] = flat[off:off + k].view(grads[n].shape)
                off += k
        arena = self._arena
        self._reset()
        self.last_arena = (arena.data_ptr(), arena.numel() * 4) if arena is not None else None


def attach(module, process_group=None, sync_bn: bool = True, bucket_bytes: int = 32 << 20,
           broadcast: bool = True, force: Optional[bool] = None) -> DistContext:
    """Make `module` (a ResNetSeg* drop-in) data-parallel: the equivalent of wrapping the
    reference network in DDP + SyncBatchNorm (train.py:100-104).  Do NOT also wrap it in
    DistributedDataParallel / convert_sync_batchnorm."""
    ctx = DistContext(process_group, sync_bn, bucket_bytes, force=force)
    ctx.bind(list(module.named_parameters()))
    module._dist = ctx if ctx.active else None
    if broadcast and ctx.world > 1:
        broadcast_parameters(module, process_group)
    return ctx


def broadcast_parameters(module, process_group=None, src: int = 0):
    """C5: rank-`src` parameters and buffers to every rank, one coalesced broadcast per dtype."""
    tensors = [p.data for p in module.parameters()] + [b for b in module.buffers()]
    by_dtype: Dict[torch.dtype, List[Tensor]] = {}
    for t in tensors:
        by_dtype.setdefault(t.dtype, []).append(t)
    for ts in by_dtype.values():
        flat = torch.cat([t.reshape(-1) for t in ts])
        dist.broadcast(flat, src=src, group=process_group)
        off = 0
        for t in ts:
            t.copy_(flat[off:off + t.numel()].view_as(t))
            off += t.numel()
    # written through p.data: no version counter moved, so invalidate the packed-weight cache of no_grad forwards
    from . import ops
    ops.weights_changed()
