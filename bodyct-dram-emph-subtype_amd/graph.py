"""hipGraph capture of a whole train step (forward + loss + backward + fused Adam update).

Small volumes make the step launch-bound: BASELINE configs[0] (ResNet-34, 1x64x128x128) issues ~700 kernels of
5-40 us each.  Everything the engine launches is a plain kernel on torch's current stream with host-side
arguments that do not change from step to step, and ``FusedAdam(capturable=True)`` keeps lr / step / bias
corrections in device memory, so the step can be captured once and replayed:

    opt = FusedAdam(module.parameters(), lr=1e-4, capturable=True)
    step = GraphedTrainStep(module, opt, lambda image, lung, cle, pse: loss_of(module(image, lung), cle, pse),
                            (image, lung, cle, pse))
    for batch in loader:
        loss = step(*batch)              # copies the batch into the static buffers, replays the graph
    scheduler.step()                     # lr changes reach the captured update through the device-side copy

BN running statistics, num_batches_tracked and the Adam state are updated by the replay exactly as by the eager step
(tests/test_models_gpu.py).

Data parallel (``distributed.attach``): under RCCL the context's collectives are plain stream work -- ``ncclAllReduce``
on the capturing stream for the SyncBN statistics, on a forked and re-joined communication stream for the gradient
buckets (rccl.py) -- so the step is captured WITH them; an eager data-parallel ResNet-50 bf16 step is host-bound
(~19 ms of launch issue for 16 ms of kernels), the replay is not.  If the runtime refuses the capture, a note goes to
stderr and the step runs eagerly; a ``gloo`` group (host-side collectives) is never captured.
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch

from . import ops
from .optim import FusedAdam


class GraphedTrainStep:
    def __init__(self, module: torch.nn.Module, optimizer: FusedAdam, loss_fn: Callable[..., torch.Tensor],
                 example_inputs: Sequence[torch.Tensor], warmup: int = 2, streams: int = 1):
        """streams = 2: the captured step keeps the eager step's second branch (weight-gradient kernels and the per-step
        weight packing fork from the capturing stream and rejoin it) -- worth it where the two chains use different pipes
        (fp32 configs: matrix-bound weight-gradient GEMMs beside the HBM-bound transforms of the data-gradient chain)."""
        if not isinstance(optimizer, FusedAdam) or not optimizer.capturable:
            raise TypeError("GraphedTrainStep needs FusedAdam(..., capturable=True)")
        ctx = getattr(getattr(module, "model", module), "_dist", None)
        if ctx is not None and not ctx.capturable:
            raise NotImplementedError("graph capture of a data-parallel step needs the RCCL transport (a gloo process "
                                      "group runs its collectives on the host)")
        if streams not in (1, 2):
            raise ValueError("GraphedTrainStep: streams must be 1 or 2")
        self.module, self.optimizer, self.loss_fn = module, optimizer, loss_fn
        engine = getattr(getattr(module, "model", module), "_engine", None)
        self.static = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                 # warm-up on a side stream, as torch.cuda.graph asks
            for _ in range(max(1, warmup)):           # (creates optimizer state, scratch buffers, plans)
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        self._keep: list = []                         # scratch buffers / device work lists the captured launches point at
        prev_streams = engine.graph_streams if engine is not None else 1
        if engine is not None:
            engine.graph_streams = streams
        try:
            with ops.capture_keepalive(self._keep), torch.cuda.graph(self.graph):
                self.static_loss = self._eager()
        except RuntimeError as exc:
            if engine is not None:
                engine.graph_streams = prev_streams
            if ctx is None:
                raise
            # collectives the runtime would not record: the data-parallel step stays correct, eagerly
            import sys
            print(f"GraphedTrainStep: capture of the data-parallel step was refused ({str(exc).splitlines()[0]}); "
                  "running eager steps", file=sys.stderr)
            self.graph = None
            self._keep.clear()
            torch.cuda.synchronize()
        if engine is not None:
            engine.graph_streams = prev_streams
        self.steps_captured_eagerly = max(1, warmup)          # the capture pass itself does not execute

    def _eager(self):
        self.optimizer.zero_grad(set_to_none=True)
        loss = self.loss_fn(*self.static)
        loss.backward()
        self.optimizer.step()
        return loss.detach()

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        if len(inputs) != len(self.static):
            raise ValueError(f"expected {len(self.static)} inputs")
        for s, t in zip(self.static, inputs):
            if s.shape != t.shape or s.dtype != t.dtype:
                raise ValueError("GraphedTrainStep: input shapes / dtypes are fixed at capture time")
            if s.data_ptr() != t.data_ptr():
                s.copy_(t)
        if self.graph is None:
            return self._eager()
        self.optimizer.sync_hyper()                   # lr moved by the scheduler since the last replay?
        self.graph.replay()
        self.optimizer.note_replayed_step()
        return self.static_loss
