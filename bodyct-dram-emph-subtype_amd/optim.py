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


def build_tables(ptrs: List[tuple], chunk: int = _lib.OPT_CHUNK):
    """Host-side layout of the kernel's work list: one DramTensorRef per tensor and one
    DramChunkRef per `chunk` elements.  ptrs: [(p, g, m, v, n), ...] as integers."""
    table = np.zeros(len(ptrs), dtype=_TABLE_DT)
    chunks = []
    for i, (p, g, m, v, n) in enumerate(ptrs):
        table[i] = (p, g, m, v, n)
        for off in range(0, n, chunk):
            chunks.append((i, 0, off))
    return table, np.array(chunks, dtype=_CHUNK_DT)


class _FusedBase(torch.optim.Optimizer):
    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._cache_key = None
        self._cache = None
        self.grad_scale = 1.0   # DDP folds the 1/world_size of the gradient mean in here

    def _tables(self, key, ptrs, device):
        if key != self._cache_key:
            table, chunks = build_tables(ptrs)
            t = torch.from_numpy(table.view(np.uint8).copy()).to(device)
            c = torch.from_numpy(chunks.view(np.uint8).copy()).to(device)
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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0 or eps < 0 or not 0 <= betas[0] < 1 or not 0 <= betas[1] < 1 or weight_decay < 0:
            raise ValueError("invalid Adam hyper-parameter")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
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
