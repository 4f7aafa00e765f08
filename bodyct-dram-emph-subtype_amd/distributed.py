"""Data parallelism for the Med3D engine: one process per GPU, RCCL over xGMI.

Stands in for what the reference gets from Lightning's ``DDPStrategy`` +
``sync_batchnorm=True`` (reference train.py:70,100-104):

  C1  gradient mean over ranks     -> bucketed, *asynchronous* all-reduce launched from
                                      inside the engine's backward as soon as a layer's
                                      gradients exist, overlapped with the rest of backward
  C2  SyncBN forward statistics    -> one all-reduce of [sum, sum^2] (2C doubles)
  C3  SyncBN backward sums         -> one all-reduce of [sum g, sum g*xhat] (2C doubles)
  C4  per-step buffer broadcast    -> dropped: SyncBN keeps running stats identical
  C5  initial parameter broadcast  -> broadcast_parameters()

``torch.distributed`` with backend ``nccl`` IS RCCL on ROCm; the same code runs on
``gloo`` (CPU or device tensors) for tests.  The engine's fused conv-BN units are invisible
to ``SyncBatchNorm.convert_sync_batchnorm``, hence this module.
"""
from __future__ import annotations

from typing import Dict, List, Optional

import torch
import torch.distributed as dist

Tensor = torch.Tensor


class DistContext:
    def __init__(self, process_group=None, sync_bn: bool = True, bucket_bytes: int = 32 << 20,
                 average: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.pg = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)
        self.sync_bn = sync_bn
        self.bucket_bytes = bucket_bytes
        self.average = average
        self._backend = dist.get_backend(process_group)
        self._pending: List[str] = []
        self._pending_bytes = 0
        self._inflight = []   # (work, flat, names, shapes)
        self._done = set()

    # ---------------------------------------------------------------- SyncBN
    def sync_bn_stats(self, sums: Tensor, count: float):
        """sums [2,C] float64 local -> (global sums, global count)."""
        if not self.sync_bn or self.world == 1:
            return sums, count
        out = sums.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.pg)
        # every rank contributes the same per-rank count (same local batch shape), so the
        # global count is known on the host without a device->host sync
        return out, count * self.world

    def all_reduce_sum(self, t: Tensor) -> Tensor:
        if not self.sync_bn or self.world == 1:
            return t
        out = t.clone()
        dist.all_reduce(out, op=dist.ReduceOp.SUM, group=self.pg)
        return out

    # ---------------------------------------------------------------- gradients
    def grads_ready(self, grads: Dict[str, Tensor], names: List[str]):
        """Called by the engine's backward when `names` have their final local gradients."""
        if self.world == 1:
            return
        for n in names:
            if n in self._done or n not in grads:
                continue
            self._done.add(n)
            self._pending.append(n)
            self._pending_bytes += grads[n].numel() * grads[n].element_size()
        if self._pending_bytes >= self.bucket_bytes:
            self._launch(grads)

    def _launch(self, grads: Dict[str, Tensor]):
        if not self._pending:
            return
        names = self._pending
        self._pending, self._pending_bytes = [], 0
        flat = torch.cat([grads[n].reshape(-1) for n in names])
        if self.average and self._backend == "nccl":
            work = dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.pg, async_op=True)
            scaled = True
        else:
            work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            scaled = False
        self._inflight.append((work, flat, names, [grads[n].shape for n in names], scaled))

    def finish(self, grads: Dict[str, Tensor]):
        """Flush, wait for every bucket, and re-point grads at the reduced buffers."""
        if self.world == 1:
            return
        self._launch(grads)
        for work, flat, names, shapes, scaled in self._inflight:
            work.wait()
            if self.average and not scaled:
                flat.div_(self.world)
            off = 0
            for n, shp in zip(names, shapes):
                k = 1
                for s in shp:
                    k *= s
                grads[n] = flat[off:off + k].view(shp)
                off += k
        self._inflight = []
        self._done = set()


def attach(module, process_group=None, sync_bn: bool = True, bucket_bytes: int = 32 << 20,
           broadcast: bool = True) -> DistContext:
    """Make `module` (a ResNetSeg* drop-in) data-parallel: the equivalent of wrapping the
    reference network in DDP + SyncBatchNorm (train.py:100-104)."""
    ctx = DistContext(process_group, sync_bn, bucket_bytes)
    module._dist = ctx if ctx.world > 1 else None
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
