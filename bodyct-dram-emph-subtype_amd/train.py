"""Trainer harness with the reference's ``train.py`` surface, without Lightning.

Mirrors reference train.py:20-108: same flag names/defaults, experiment layout
``<model_path>/subtyping_<arch>/checkpoints``, newest-checkpoint discovery (or ``--ckp``),
``--reload_only_weights`` (greedy weight load + fresh optimizer, train.py:83-89) vs full resume,
one checkpoint per epoch named like Lightning's ``'{epoch:02d}'`` (``epoch=NN.ckpt``, train.py:92-99),
Adam + ExponentialLR(0.95) per epoch (models.py:685-698), one process per GPU under
``torch.distributed.run`` with RCCL (``distributed.attach`` = DDP + SyncBatchNorm,
train.py:70,100-104).  Checkpoints are Lightning-1.9-shaped dicts (``state_dict`` with the
``model.`` prefix of the LightningModule attribute, ``optimizer_states``, ``lr_schedulers``,
``epoch``, ``global_step``) so they interoperate with the reference's loaders
(test.py:66-72, processor.py:85-87).

The reference's DataModule (SimpleITK / COPDGene cache) is out of scope; ``--synthetic``
(default) feeds batches with the reference's batch contract (models.py:541-548).
"""
from __future__ import annotations

if not __package__:          # imported top-level (this directory on sys.path): bind to the package, see _dropin.py
    import _dropin
    __package__ = _dropin.adopt(__name__)

import glob
import logging
import os
from argparse import ArgumentParser
from pathlib import Path

import torch

from . import models
from .transforms import TrainAugment
from .utils import load_state_dict_greedy


def build_parser() -> ArgumentParser:
    p = ArgumentParser()
    p.add_argument("--model_arch", default="med3ddram50", type=str)
    p.add_argument("--lr", "--learning-rate", default=0.0001, type=float)
    p.add_argument("--ngpus", default=1, type=int)
    p.add_argument("--momentum", default=0.9, type=float)
    p.add_argument("--reload_only_weights", default=1, type=int)
    p.add_argument("--weight_decay", default=1e-5, type=float)
    p.add_argument("--ckp", type=str, default=None)
    p.add_argument("--target_size", default=(128, 224, 288), type=int, nargs=3)
    p.add_argument("--data_path", default="", type=str)
    p.add_argument("--train_csv", default="", type=str)
    p.add_argument("--valid_csv", default="", type=str)
    p.add_argument("--test_csv", default="", type=str)
    p.add_argument("--model_path", default="./models/", type=str)
    p.add_argument("--workers", default=2, type=int)
    p.add_argument("--batch_size", default=1, type=int)
    p.add_argument("--num_samples", default=128, type=int)
    p.add_argument("--local_rank", default=0, type=int, help="this argument is not used and should be ignored")
    p.add_argument("--max_epochs", default=120, type=int)          # Trainer flag the reference sets (train.py:49)
    p.add_argument("--log_every_n_steps", default=5, type=int)
    p.add_argument("--synthetic", default=1, type=int)
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--augment", default=0, type=int, help="1: GPU-side train-time augmentations (models.py:66-74)")
    # Lightning Trainer flag the reference exposes through Trainer.add_argparse_args (train.py:46): "bf16" runs the
    # step under autocast(bfloat16) there, the bf16 storage path here
    p.add_argument("--precision", default="32", type=str, choices=("32", "bf16"))
    return p


class SyntheticSubtypeData:
    """Batches with the reference's contract: image f32 [B,D,H,W] (windowed + z-scored),
    lung_mask / em_mask bool, cls_label 0-5, pse_label 0-2, index (models.py:541-548)."""

    def __init__(self, num_samples, batch_size, size, rank=0, world=1, device="cuda", seed=0):
        self.n = max(1, num_samples // (batch_size * world))
        self.B, self.size, self.rank, self.device, self.seed = batch_size, tuple(size), rank, device, seed

    def __len__(self):
        return self.n

    def epoch(self, epoch):
        D, H, W = self.size
        g = torch.Generator(device=self.device).manual_seed(self.seed + 1000 * epoch + self.rank)
        gl = torch.Generator().manual_seed(self.seed + 1000 * epoch + self.rank)
        z = (torch.arange(D, device=self.device).float() - (D - 1) / 2) / (0.4 * D)
        y = (torch.arange(H, device=self.device).float() - (H - 1) / 2) / (0.35 * H)
        x = (torch.arange(W, device=self.device).float() - (W - 1) / 2) / (0.4 * W)
        lung = (z[:, None, None] ** 2 + y[None, :, None] ** 2 + x[None, None, :] ** 2) <= 1.0
        for i in range(self.n):
            image = torch.randn(self.B, D, H, W, device=self.device, generator=g)
            lm = lung[None].expand(self.B, D, H, W)
            yield {"image": image, "lung_mask": lm, "em_mask": (image < -1.0) & lm,
                   "cls_label": torch.randint(0, 6, (self.B,), generator=gl).to(self.device),
                   "pse_label": torch.randint(0, 3, (self.B,), generator=gl).to(self.device),
                   "index": (torch.arange(self.B) + i * self.B).unsqueeze(-1).to(self.device)}


def augment_batch(batch, augment):
    """apply one parameter draw per sample to the image and its masks (the reference's dict transforms run per
    sample in the DataLoader workers)"""
    keys = [k for k in batch if k == "image" or k.endswith("_mask")]
    cols = {k: [] for k in keys}
    for b in range(batch["image"].shape[0]):
        out = augment({k: batch[k][b] for k in keys})
        for k in keys:
            cols[k].append(out[k])
    new = dict(batch)
    new.update({k: torch.stack(v) for k, v in cols.items()})
    return new


# ------------------------------------------------------------------ checkpoints (Lightning-1.9 shape)
def checkpoint_dict(module, optimizer, scheduler, epoch, global_step, args) -> dict:
    return {"epoch": epoch, "global_step": global_step, "pytorch-lightning_version": "1.9.1",
            "state_dict": {k: v.detach().cpu() for k, v in module.state_dict().items()},   # keys 'model.<med3d key>'
            "optimizer_states": [optimizer.state_dict()], "lr_schedulers": [scheduler.state_dict()],
            "hyper_parameters": {"args": dict(vars(args))}}


def find_checkpoint(ckp_dir: Path, ckp: str = None):
    """train.py:77-82: explicit --ckp, else the newest *.ckpt / *.pth by ctime."""
    files = list(glob.glob(ckp_dir.as_posix() + "/*.ckpt")) + list(glob.glob(ckp_dir.as_posix() + "/*.pth"))
    if not files:
        return None
    return (ckp_dir / ckp).as_posix() if ckp is not None else max(files, key=os.path.getctime)


def restore(module, optimizer, scheduler, path, reload_only_weights: bool):
    """Returns the epoch to start from."""
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    sd = ckpt["state_dict"] if "state_dict" in ckpt else ckpt
    if reload_only_weights:
        load_state_dict_greedy(module, sd)        # train.py:83-89: weights only, fresh optimizer
        return 0
    module.load_state_dict(sd)
    optimizer.load_state_dict(ckpt["optimizer_states"][0])
    scheduler.load_state_dict(ckpt["lr_schedulers"][0])
    return int(ckpt["epoch"]) + 1


def run_training_job(argv=None):
    args = build_parser().parse_args(argv)
    args.exp_name = f"subtyping_{args.model_arch}"
    exp_path = Path(args.model_path) / args.exp_name
    ckp_path = exp_path / "checkpoints"
    ckp_path.mkdir(exist_ok=True, parents=True)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    logging.basicConfig(level=logging.INFO, format="%(asctime)s [%(levelname)s] %(message)s",
                        handlers=[logging.FileHandler(f"{exp_path}/debug.log"), logging.StreamHandler()])
    if not torch.cuda.is_available():
        raise SystemExit("training runs on MI355X only (no CPU fallback)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)   # RCCL (reference: gloo/nccl)
    torch.manual_seed(args.seed)
    module = (models.ScanRegLightningModule if "dram" in args.model_arch else models.ScanCLSLightningModule)(args)
    module = module.to(device)
    if args.precision == "bf16":
        module.model.storage_dtype = torch.bfloat16
    (optimizer,), (scheduler,) = module.configure_optimizers()
    start_epoch = 0
    found = find_checkpoint(ckp_path, args.ckp)
    if found is not None:
        start_epoch = restore(module, optimizer, scheduler, found, bool(args.reload_only_weights))
        logging.info(f"restored {found} (weights only: {bool(args.reload_only_weights)}) -> epoch {start_epoch}")
    if world > 1:
        from . import distributed as ddist
        ddist.attach(module.model)                 # DDP + SyncBatchNorm semantics (train.py:100-104)
    data = SyntheticSubtypeData(args.num_samples, args.batch_size, args.target_size, rank, world, device, args.seed)
    val_data = SyntheticSubtypeData(max(args.batch_size * world, args.num_samples // 4), args.batch_size, args.target_size,
                                    rank, world, device, args.seed + 7)
    augment = TrainAugment() if getattr(args, "augment", 0) else None        # models.py:66-74 (train mode only)
    global_step = 0
    best = (float("inf"), None)
    for epoch in range(start_epoch, args.max_epochs):
        module.train()
        running, step_outputs, losses = 0.0, [], []
        for i, batch in enumerate(data.epoch(epoch)):
            if augment is not None:
                batch = augment_batch(batch, augment)
            optimizer.zero_grad(set_to_none=True)
            out = module.training_step(batch, i)
            out["loss"].backward()
            optimizer.step()
            losses.append(out["loss"].detach())
            step_outputs.append({k: v for k, v in out.items() if k != "loss"})
            global_step += 1
            if global_step % args.log_every_n_steps == 0:
                running = float(out["loss"])
                if rank == 0:
                    logging.info(f"epoch {epoch} step {global_step} train_loss {running:.5f} "
                                 f"lr {optimizer.param_groups[0]['lr']:.3e}")
        # Lightning's fit loop: validation epoch, then the epoch-end hooks (gather + de-dup + class-weight update,
        # models.py:287-317 / :367-379), then the scheduler and ModelCheckpoint
        module.eval()
        val_outputs = [module.validation_step(b, i) for i, b in enumerate(val_data.epoch(epoch))]
        ev = module.validation_epoch_end(val_outputs)
        et = module.training_epoch_end(step_outputs)
        train_loss = float(torch.stack(losses).mean())
        if rank == 0:
            logging.info(f"epoch {epoch}: train_loss {train_loss:.5f} train acc cle/pse {float(et['acc_cle']):.3f}/"
                         f"{float(et['acc_pse']):.3f} val acc cle/pse {float(ev['acc_cle']):.3f}/{float(ev['acc_pse']):.3f} "
                         f"class weights {module.cle_class_weights.tolist()}")
        scheduler.step()                           # ExponentialLR(gamma=0.95), per epoch
        if rank == 0:                              # ModelCheckpoint(save_top_k=-1, every_n_epochs=1, '{epoch:02d}')
            path = ckp_path / f"epoch={epoch:02d}.ckpt"
            torch.save(checkpoint_dict(module, optimizer, scheduler, epoch, global_step, args), path)
            if train_loss < best[0]:               # monitor='train_loss' (train.py:92-99)
                best = (train_loss, path)
    best_path = best[1]
    if world > 1:
        # trainer.test(ckpt_path='best') restores the SAME checkpoint on every rank: rank 0 tracked `best`, so its
        # choice is broadcast, and nobody reads the file before rank 0 has finished writing it
        box = [str(best_path) if best_path is not None else None]
        torch.distributed.broadcast_object_list(box, src=0)
        best_path = box[0]
        torch.distributed.barrier()
    if best_path is not None:                      # trainer.test(ckpt_path='best') (train.py:108)
        module.load_state_dict(torch.load(best_path, map_location="cpu", weights_only=False)["state_dict"])
        from . import ops
        ops.weights_changed()
        module.eval()
        test_outputs = [module.test_step(b, i) for i, b in enumerate(val_data.epoch(10_000))]
        te = module.test_epoch_end(test_outputs)
        if rank == 0:
            logging.info(f"test (best = {best_path}): acc cle/pse {float(te['acc_cle']):.3f}/{float(te['acc_pse']):.3f}")
    if world > 1:
        torch.distributed.destroy_process_group()
    return module


if __name__ == "__main__":
    print("Running training job.")
    run_training_job()
