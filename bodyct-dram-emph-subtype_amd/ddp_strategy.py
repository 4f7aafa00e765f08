"""Binding of the data-parallel path to PyTorch-Lightning's strategy seam.

The reference builds ``DDPStrategy(process_group_backend=..., find_unused_parameters=False)`` and passes
``sync_batchnorm=True`` to the Trainer (reference train.py:70,100-104).  Lightning 1.9 then
  1. initialises the process group (backend "nccl" IS RCCL on ROCm),
  2. converts ``_BatchNorm`` modules to SyncBatchNorm (harmless here: the BatchNorm3d modules of the drop-in
     network are parameter containers; their forward is never called),
  3. wraps the LightningModule in ``DistributedDataParallel`` (``DDPStrategy._setup_model``).
Step 3 must NOT happen for this engine: its gradients are produced by one fused backward that launches the
RCCL all-reduces itself (``distributed.attach``), and DDP's reducer hooks would all-reduce them a second
time.  ``make_ddp_strategy()`` returns a ``DDPStrategy`` whose ``_setup_model`` attaches the engine's
``DistContext`` to ``module.model`` and hands the module back unwrapped.

    strategy = make_ddp_strategy(process_group_backend="nccl")
    trainer = pl.Trainer.from_argparse_args(args, strategy=strategy, sync_batchnorm=True, devices=args.ngpus, ...)

pytorch_lightning is not installed in the build image; the class is created lazily from whatever
``pytorch_lightning.strategies.DDPStrategy`` is importable (tests substitute a stand-in with the 1.9 hook names).
"""
from __future__ import annotations

from . import distributed as ddist


def make_ddp_strategy(bucket_bytes: int = 32 << 20, sync_bn: bool = True, **ddp_kwargs):
    try:
        from pytorch_lightning.strategies import DDPStrategy
    except Exception as e:  # noqa: BLE001
        raise RuntimeError("make_ddp_strategy needs pytorch_lightning (reference pins 1.9.1); without Lightning use "
                           "distributed.attach(module.model) directly, as bodyct-dram-emph-subtype_amd/train.py does") from e

    class DramDDPStrategy(DDPStrategy):
        """DDPStrategy that leaves the module unwrapped and lets the HIP engine run the collectives."""

        def _setup_model(self, model):                       # Lightning 1.9: returns DistributedDataParallel(model)
            lm = getattr(model, "module", model)             # _LightningModuleWrapperBase(pl_module) or the module itself
            lm = getattr(lm, "_forward_module", lm)
            net = getattr(lm, "model", None)
            if net is None or not hasattr(net, "_engine"):
                raise RuntimeError("DramDDPStrategy: the LightningModule must hold the drop-in network as `.model` "
                                   "(reference models.py:171, :408)")
            self.dram_context = ddist.attach(net, sync_bn=sync_bn, bucket_bytes=bucket_bytes)
            return model

        def _register_ddp_hooks(self):                       # no DistributedDataParallel instance to hook
            return None

    ddp_kwargs.setdefault("find_unused_parameters", False)
    return DramDDPStrategy(**ddp_kwargs)
