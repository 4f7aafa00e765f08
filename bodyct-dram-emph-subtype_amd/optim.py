"""Fused multi-tensor optimizers on libdram_hip (``dram_adam_multi`` / ``dram_sgd_multi``).

Drop-in for ``torch.optim.Adam(self.parameters(), lr=args.lr)`` at reference
models.py:385-387 / :689-691 (and the SGD(momentum, weight_decay) variant whose
arguments the reference keeps, train.py:25,27): ``torch.optim.Optimizer`` subclasses, so
``ExponentialLR`` (models.py:392-394), ``param_groups[0]['lr']``, ``state_dict()`` /
``load_state_dict()`` (keys ``step``, ``exp_avg``, ``exp_avg_sq``) and Lightning's
``step(closure)`` protocol keep working.  One kernel launch updates every parameter.
"""
from __future__ import annotations

import ctypes
from typing import List

import numpy as np
import torch

from . import _lib, ops

_TABLE_DT = np.dtype([("p", "<u8"), ("g", "<u8"), ("m", "<u8"), ("v", "<u8"), ("n", "<i8")])
_CHUNK_DT = np.dtype([("tensor", "<i4"), ("pad", "<i4"), ("offset", "<i8")])
assert _TABLE_DT.itemsize == ctypes.sizeof(_lib.DramTensorRef)
assert _CHUNK_DT.itemsize == ctypes.sizeof(_lib.DramChunkRef)


def build_chunks(sizes, chunk: int = _lib.OPT_CHUNK):
    """One DramChunkRef per `chunk` elements of every tensor (depends on the tensor SIZES only)."""
    parts = []
    for i, n in enumerate(sizes):
        c = np.zeros((n + chunk - 1) // chunk, dtype=_CHUNK_DT)
        c["tensor"] = i
        c["offset"] = np.arange(0, n, chunk, dtype=np.int64)
        parts.append(c)
    return np.concatenate(parts) if parts else np.zeros(0, dtype=_CHUNK_DT)


def build_tables(ptrs: List[tuple], chunk: int = _lib.OPT_CHUNK):
    """Host-side layout of the kernel's work list: one DramTensorRef per tensor and one
    DramChunkRef per `chunk` elements.  ptrs: [(p, g, m, v, n), ...] as integers."""
    table = np.array(ptrs, dtype=_TABLE_DT) if ptrs else np.zeros(0, dtype=_TABLE_DT)
    return table, build_chunks([t[4] for t in ptrs], chunk)


class _FusedBase(torch.optim.Optimizer):
    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._cache_key = None
        self._cache = None
        self._chunk_cache = None      # (sizes, device) -> (device chunk list, count): independent of the pointers
        self._pinned_spare = None     # pinned staging pair for the next graph capture (capturable mode)
        self._pinned_owned = []       # pairs captured graphs read on every replay: never written again
        self.grad_scale = 1.0   # DDP folds the 1/world_size of the gradient mean in here

    def _tables(self, key, ptrs, device):
        if key != self._cache_key:
            table, chunks = build_tables(ptrs)
            ht, hc = torch.from_numpy(table.view(np.uint8).copy()), torch.from_numpy(chunks.view(np.uint8).copy())
            capturing = torch.cuda.is_current_stream_capturing()
            if capturing:
                # inside a hipGraph capture neither a pageable host->device copy nor a pinned allocation is
                # allowed: the tables go through a pinned pair that an EAGER step allocated beforehand (sizes depend
                # only on the parameter sizes).  The pair now belongs to this graph -- its copy node re-reads it on
                # every replay -- so it is never written again; the next eager step allocates a new spare.
                sp = self._pinned_spare
                if sp is None or sp[0].numel() != ht.numel() or sp[1].numel() != hc.numel():
                    raise RuntimeError("run one eager step with the same parameters before capturing")
                self._pinned_spare = None
                sp[0].copy_(ht)
                sp[1].copy_(hc)
                t = torch.empty(ht.shape, dtype=ht.dtype, device=device)
                c = torch.empty(hc.shape, dtype=hc.dtype, device=device)
                t.copy_(sp[0], non_blocking=True)
                c.copy_(sp[1], non_blocking=True)
                self._pinned_owned.append(sp)
                ops._keep_for_capture((t, c))         # the captured update reads them on every replay
            else:
                # Gradients are fresh tensors every step (zero_grad(set_to_none=True)), so the POINTER table changes
                # every step.  A pageable host->device copy blocks the host until the stream has drained -- the host then
                # starts issuing the next step on an idle GPU, every step -- so the table goes through pinned memory,
                # asynchronously; the chunk list depends on the sizes only and is uploaded once.
                ck = (tuple(int(n) for n in table["n"]), str(device))
                if self._chunk_cache is None or self._chunk_cache[0] != ck:
                    self._chunk_cache = (ck, hc.to(device), len(chunks))
                c = self._chunk_cache[1]
                # (a fresh pinned block per step from torch's caching host allocator, which hands a block out again
                # only after the copy that read it has executed: the host runs several steps ahead of the GPU, a ring
                # of two staging buffers with an event wait measured 13 ms of host stall per step)
                t = torch.empty(ht.shape, dtype=ht.dtype, device=device)
                t.copy_(ht.pin_memory(), non_blocking=True)
                if getattr(self, "capturable", False) and self._pinned_spare is None:
                    self._pinned_spare = (torch.empty_like(ht).pin_memory(), torch.empty_like(hc).pin_memory())
            self._cache_key, self._cache = key, (t, c, len(chunks))
        return self._cache

    @staticmethod
    def _check(p):
        if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
            raise RuntimeError("fused optimizer: parameters must be contiguous fp32 device tensors")
        g = p.grad
        if g.dtype != torch.float32 or not g.is_contiguous() or g.is_sparse:
            raise RuntimeError("fused optimizer: gradients must be dense contiguous fp32")


class FusedAdam(_FusedBase):
    """capturable=True keeps lr / betas / eps / weight decay / grad scale AND the step count in a device tensor
    (`dram_adam_multi_dev`): the update can then be captured in a hipGraph together with forward and backward
    (graph.GraphedTrainStep) and replayed across steps and lr changes; requires one parameter group whose
    parameters share their step count."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, capturable=False):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.capturable = bool(capturable)
        self._hyper = None            # device float32[7]: lr, b1, b2, eps, wd, grad_scale, step (int32 bits)
        self._hyper_host = None
        self._replayed = 0            # graph replays whose step the host-side state has not absorbed yet

    # ---- capturable mode -----------------------------------------------------------------------------------
    def _host_hyper(self, group):
        b1, b2 = group["betas"]
        return [float(group["lr"]), float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                float(self.grad_scale)]

    def sync_hyper(self):
        """Push lr (ExponentialLR mutates param_groups[0]['lr']) and friends to the device copy when they changed.
        Called by step() and by GraphedTrainStep before every replay; never inside a capture."""
        if self._hyper is None:
            return
        want = self._host_hyper(self.param_groups[0])
        if want != self._hyper_host:
            self._hyper[:6].copy_(torch.tensor(want, dtype=torch.float32))
            self._hyper_host = want

    def note_replayed_step(self):
        """A captured step was replayed: the device-side step counter advanced without this object's step(), and
        the replayed update rewrote every parameter through raw pointers -- no torch version counter moved -- so
        the packed / Winograd-transformed weights cached for no_grad forwards are stale from here on."""
        self._replayed += 1
        ops.weights_changed()

    def _absorb_replays(self):
        if self._replayed:
            for st in self.state.values():
                if "step" in st:
                    st["step"] += self._replayed
            self._replayed = 0

    def state_dict(self):
        self._absorb_replays()
        return super().state_dict()

    def _step_capturable(self):
        if len(self.param_groups) != 1:
            raise RuntimeError("FusedAdam(capturable=True) supports one parameter group")
        group = self.param_groups[0]
        plist = [p for p in group["params"] if p.grad is not None]
        if not plist:
            return
        capturing = torch.cuda.is_current_stream_capturing()
        self._absorb_replays()
        steps = set()
        for p in plist:
            self._check(p)
            st = self.state[p]
            if len(st) == 0:
                if capturing:
                    raise RuntimeError("run at least one eager step before capturing (optimizer state is created lazily)")
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            steps.add(int(st["step"].item()))
        if len(steps) != 1:
            raise RuntimeError("FusedAdam(capturable=True): parameters must share their step count")
        if self._hyper is None:
            if capturing:
                raise RuntimeError("run at least one eager step before capturing")
            self._hyper_host = self._host_hyper(group)
            self._hyper = torch.tensor(self._hyper_host + [0.0], dtype=torch.float32, device=plist[0].device)
            self._hyper[6:].view(torch.int32).fill_(int(steps.pop()))      # the kernel keeps the step count as int32 bits
        elif not capturing:
            self.sync_hyper()
        ptrs = [(p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                 self.state[p]["exp_avg_sq"].data_ptr(), p.numel()) for p in plist]
        t, c, n = self._tables((0, "dev", tuple(ptrs)), ptrs, plist[0].device)
        ops.adam_multi_dev(t, c, n, self._hyper)
        if not capturing:             # a capture pass records the update without executing it; replays are counted
            for p in plist:           # through note_replayed_step()
                self.state[p]["step"] += 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        if self.capturable:
            self._step_capturable()
            ops.weights_changed()
            return loss
        for gi, group in enumerate(self.param_groups):
            by_step = {}
            for p in group["params"]:
                if p.grad is None:
                    continue
                self._check(p)
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"].item()), []).append(p)
            b1, b2 = group["betas"]
            for step, plist in by_step.items():
                ptrs = [(p.data_ptr(), p.grad.data_ptr(), self.state[p]["exp_avg"].data_ptr(),
                         self.state[p]["exp_avg_sq"].data_ptr(), p.numel()) for p in plist]
                key = (gi, tuple(ptrs))
                t, c, n = self._tables(key, ptrs, plist[0].device)
                ops.adam_multi(t, c, n, float(group["lr"]), b1, b2, group["eps"], group["weight_decay"],
                               1.0 - b1 ** step, 1.0 - b2 ** step, float(self.grad_scale))
        ops.weights_changed()        # the kernel wrote the parameters through raw pointers
        return loss


class FusedSGD(_FusedBase):
    def __init__(self, params, lr=1e-3, momentum=0.0, weight_decay=0.0):
        if lr < 0 or momentum < 0 or weight_decay < 0:
            raise ValueError("invalid SGD hyper-parameter")
        super().__init__(params, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            first, rest = [], []
            for p in group["params"]:
                if p.grad is None:
                    continue
                self._check(p)
                st = self.state[p]
                if group["momentum"] != 0 and "momentum_buffer" not in st:
                    st["momentum_buffer"] = torch.zeros_like(p)
                    first.append(p)
                else:
                    rest.append(p)
            for is_first, plist in ((True, first), (False, rest)):
                if not plist:
                    continue
                ptrs = [(p.data_ptr(), p.grad.data_ptr(),
                         self.state[p]["momentum_buffer"].data_ptr() if group["momentum"] != 0 else 0, 0, p.numel())
                        for p in plist]
                t, c, n = self._tables((gi, is_first, tuple(ptrs)), ptrs, plist[0].device)
                ops.sgd_multi(t, c, n, float(group["lr"]), group["momentum"], group["weight_decay"], is_first,
                              float(self.grad_scale))
        ops.weights_changed()
        return loss
