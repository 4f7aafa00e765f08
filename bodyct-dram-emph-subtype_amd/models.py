"""Host-side mirror of the reference's task modules (reference models.py:160-698).

``ScanCLSLightningModule`` / ``ScanRegLightningModule`` keep the reference's method
surface for the train / predict path -- ``forward``, ``training_step``, ``shared_step``,
``predict_step``, ``configure_optimizers`` -- on top of the HIP engine.  When
``pytorch_lightning`` is importable they subclass ``pl.LightningModule`` (so
``Trainer.fit`` / ``processor.py`` work unchanged); otherwise a plain ``nn.Module`` with the
same methods (this image has no Lightning).  Epoch-end reporting, plotting and csv dumps
(models.py:278-379, :594-682) are out of scope.

The dRAM losses (models.py:512-531 + metrics.py) run as fused HIP kernels
(csrc/head_loss.hip) behind ``torch.autograd.Function``; O(B) scalar algebra stays in torch.
"""
from __future__ import annotations

if not __package__:          # imported top-level (this directory on sys.path): bind to the package, see _dropin.py
    import _dropin
    __package__ = _dropin.adopt(__name__)

from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .optim import FusedAdam
from .utils import get_model_by_name

try:  # pragma: no cover - not installed in the build image
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    pl = None

    class _Base(nn.Module):
        """Minimal stand-in for pl.LightningModule (same hook names)."""

        def save_hyperparameters(self, *a, **k):
            pass

        def log(self, *a, **k):
            pass

TRAIN_PHASE, VALID_PHASE, TEST_PHASE, PREDICT_PHASE = "train", "validate", "test", "predict"

# dataset.py:99-112 (COPDGeneSubtyping.cle_ratio_map / pse_ratio_map)
CLE_RATIO_MAP = {0: (0.0, 0.01), 1: (0.01, 0.05), 2: (0.05, 0.1), 3: (0.1, 0.2), 4: (0.2, 0.3), 5: (0.3, 1.0001)}
PSE_RATIO_MAP = {0: (0.0, 0.01), 1: (0.01, 0.05), 2: (0.05, 1.0001)}
BETA, GAMMA = 0.7338, 0.2578   # models.py:414-415


def _band_table(ratio_map, tightness=1.0):
    """_generate_regression_labels (models.py:495-510) as a [n_classes, 2] lookup table."""
    rows = []
    for c in sorted(ratio_map):
        lb, ub = ratio_map[c]
        if lb < 1e-7:
            rows.append((0.0, 0.0))
        else:
            m, span = (lb + ub) / 2.0, (ub - lb) * tightness / 2.0
            assert m - span < m + span
            rows.append((m - span, m + span))
    return torch.tensor(rows, dtype=torch.float32)


_CLE_BANDS, _PSE_BANDS = _band_table(CLE_RATIO_MAP), _band_table(PSE_RATIO_MAP)


_BANDS_DEV: Dict[tuple, torch.Tensor] = {}


def _bands(which: str, device) -> torch.Tensor:
    key = (which, device)
    tab = _BANDS_DEV.get(key)
    if tab is None:      # one host->device copy per device (none per step: the step stays graph-capturable)
        tab = _BANDS_DEV[key] = (_CLE_BANDS if which == "cle" else _PSE_BANDS).to(device).contiguous()
    return tab


def generate_regression_labels(cls_targets: torch.Tensor, which: str) -> torch.Tensor:
    return _bands(which, cls_targets.device)[cls_targets.long()]


def interval_regression_loss(outs, reg_targets, weight_factors):
    """models.py:512-521 (O(B) scalars: torch glue)."""
    n = torch.cat([outs.unsqueeze(1), reg_targets], dim=1)
    n = BETA * n ** GAMMA
    K = (0.5 * (n[:, 2] - n[:, 1])) ** 2
    unh = (n[:, 0] - (n[:, 2] + n[:, 1]) / 2.0) ** 2 - K
    return (10.0 * F.leaky_relu(unh, negative_slope=0.0) * weight_factors).sum()


def ratio_to_label(ratios: torch.Tensor, which: str) -> torch.Tensor:
    """_ratio_to_label (models.py:533-537) without the per-sample .item() syncs."""
    rm = CLE_RATIO_MAP if which == "cle" else PSE_RATIO_MAP
    lo = torch.tensor([rm[k][0] for k in sorted(rm)], device=ratios.device)
    hi = torch.tensor([rm[k][1] for k in sorted(rm)], device=ratios.device)
    hit = (lo[None] <= ratios[:, None]) & (ratios[:, None] < hi[None])
    return hit.float().argmax(1).long()


class _SegLossFn(torch.autograd.Function):
    """(mul_loss, seg_loss) of _segmentation_loss (models.py:523-531): dice of the
    lung-masked maps + masked, class-balanced BCE (metrics.py:10-37), labels/masks
    nearest-resized on the fly (models.py:567-570).  One HBM pass forward, one backward."""

    @staticmethod
    def forward(ctx, cle, pse, lungs, ems, binary, smooth=1e-7, smoothness=0.85):
        """smooth: BinaryDice's constant (1e-7 at models.py:412); smoothness: the in-mask BCE weight
        (0.85 at models.py:529)."""
        B, D, H, W = cle.shape
        part = ops.segloss_fwd(cle, pse, lungs, ems, binary, smoothness)
        st, A1, A0, I, S1, S2 = part.double().sum(0).unbind(0)
        N = float(B * D * H * W)
        alpha = (1.0 - st / B).clamp(0.3, 0.7)          # metrics.py:18
        sw = alpha * st + (1.0 - alpha) * (N - st)       # sum of w
        seg = (alpha * A1 + (1.0 - alpha) * A0) / sw
        den = S1 + S2 + smooth                           # BinaryDice(1e-7), models.py:412
        mul = (2.0 * I + smooth) / den
        ctx.smooth, ctx.smoothness = float(smooth), float(smoothness)
        ctx.save_for_backward(cle, pse, lungs, ems, binary, torch.stack([alpha, sw, den, I]))
        return mul.float(), seg.float()

    @staticmethod
    def backward(ctx, g_mul, g_seg):
        cle, pse, lungs, ems, binary, sc = ctx.saved_tensors
        alpha, sw, den, I = sc.unbind(0)
        gm = g_mul.double() if g_mul is not None else torch.zeros((), dtype=torch.float64, device=cle.device)
        gs = g_seg.double() if g_seg is not None else torch.zeros((), dtype=torch.float64, device=cle.device)
        z = torch.zeros((), dtype=torch.float64, device=cle.device)
        coef = torch.stack([gm * 2.0 / den, gm * (2.0 * I + ctx.smooth) / (den * den), gs * alpha / sw,
                            gs * (1.0 - alpha) / sw, z, z, z, z]).float()
        gcle, gpse = ops.segloss_bwd(cle, pse, lungs, ems, binary, coef, ctx.smoothness)
        return gcle, gpse, None, None, None, None, None


def segmentation_loss(dense_cle, dense_pse, ems, lungs, binary):
    """dense_*: [B,1,d,h,w]; ems/lungs: full-res [B,1,D,H,W] float; binary [B] float."""
    B = dense_cle.shape[0]
    c4 = dense_cle.reshape(B, *dense_cle.shape[-3:]).contiguous()
    p4 = dense_pse.reshape(B, *dense_pse.shape[-3:]).contiguous()
    l4 = lungs.reshape(B, *lungs.shape[-3:]).contiguous()
    e4 = ems.reshape(B, *ems.shape[-3:]).contiguous()
    return _SegLossFn.apply(c4, p4, l4, e4, binary.float().contiguous())


class _RegLossFn(torch.autograd.Function):
    """The whole train loss of ScanRegLightningModule.shared_step (models.py:549-574) as three launches: the seg-loss
    pass over the dense maps, one O(B) tail kernel (fold, dice/BCE, both interval losses, the total, the backward's
    coefficients), and -- in backward -- the seg-loss gradient pass.  The four components come back detached."""

    @staticmethod
    def forward(ctx, cle, pse, lungs, ems, reg_cle, reg_pse, cle_labels, pse_labels, cle_w, pse_w):
        B, D, H, W = cle.shape
        binary = torch.logical_or(cle_labels > 0, pse_labels > 0).float()            # models.py:566
        part = ops.segloss_fwd(cle, pse, lungs, ems, binary, 0.85)                    # 0.85: models.py:529
        out, coef, greg = ops.regloss_tail(part, reg_cle, reg_pse, cle_labels, pse_labels, cle_w, pse_w,
                                           _bands("cle", cle.device), _bands("pse", cle.device), B * D * H * W,
                                           1e-7, BETA, GAMMA)                         # 1e-7: BinaryDice, models.py:412
        ctx.save_for_backward(cle, pse, lungs, ems, binary, coef, greg)
        loss, lc, lp, mul, seg = out.unbind(0)
        ctx.mark_non_differentiable(lc, lp, mul, seg)
        return loss, lc, lp, mul, seg

    @staticmethod
    def backward(ctx, g, *_):
        cle, pse, lungs, ems, binary, coef, greg = ctx.saved_tensors
        gcle, gpse = ops.segloss_bwd(cle, pse, lungs, ems, binary, coef * g, 0.85)
        gr = greg * g
        return gcle, gpse, None, None, gr[0], gr[1], None, None, None, None


def reg_train_loss(dense_outs, reg_outs, lungs, ems, cle_labels, pse_labels, cle_w, pse_w):
    """Train branch of ScanRegLightningModule.shared_step (models.py:549-574): loss = interval(cle) + interval(pse)
    + 2 dice + BCE.  Same terms as interval_regression_loss / segmentation_loss above, closed in one tail kernel."""
    B = dense_outs[0].shape[0]

    def vol(t):
        return t.reshape(B, *t.shape[-3:]).float().contiguous()

    def row(t, dtype=torch.float32):
        return t.reshape(B).to(dtype).contiguous()

    loss, lc, lp, mul, seg = _RegLossFn.apply(vol(dense_outs[0]), vol(dense_outs[1]), vol(lungs), vol(ems),
                                              row(reg_outs[0]), row(reg_outs[1]), row(cle_labels, torch.int64),
                                              row(pse_labels, torch.int64), row(cle_w), row(pse_w))
    return loss, dict(loss_cle=lc, loss_pse=lp, mul_loss=mul, seg_loss=seg)


def cls_train_loss(cls_outs, cle_labels, pse_labels, cle_cw, pse_cw):
    """models.py:248-258: two class-weighted cross-entropies on [B,6] / [B,3] (K16, glue)."""
    loss_cle = F.cross_entropy(cls_outs[0], cle_labels, weight=cle_cw)
    loss_pse = F.cross_entropy(cls_outs[1], pse_labels, weight=pse_cw)
    return loss_cle + loss_pse, dict(loss_cle=loss_cle, loss_pse=loss_pse)


def update_class_weights(weights: torch.Tensor, y_true: torch.Tensor, y_pred: torch.Tensor) -> torch.Tensor:
    """models.py:367-377: w <- w * (1 - per-class accuracy), renormalised; the accuracy is diag / row sums of the
    confusion matrix over the labels that occur (sklearn.metrics.confusion_matrix semantics)."""
    labels = torch.unique(torch.cat([y_true, y_pred]))
    if labels.numel() != weights.numel():
        raise ValueError(f"class-weight update: {labels.numel()} classes occur, {weights.numel()} weights "
                         "(the reference's element-wise product fails the same way)")
    hit = (y_true[None, :] == labels[:, None])
    acc = (hit & (y_pred[None, :] == labels[:, None])).sum(1).double() / hit.sum(1).double()
    w = weights.double().to(acc.device) * (1.0 - acc)
    return (w / w.sum()).to(weights.dtype).cpu()


class _ScanModule(_Base):
    def __init__(self, args):
        self.args = args
        super().__init__()
        self.model = get_model_by_name(args.model_arch)
        self.save_hyperparameters()
        self.trace = True
        # per-class loss weights; the reference reads them from the datamodule's sampler
        # (models.py:248-252, 556-561) and rescales them each epoch (:369-379)
        self.cle_class_weights = torch.full((6,), 1.0 / 6)
        self.pse_class_weights = torch.full((3,), 1.0 / 3)

    def forward(self, x, lungs):
        return self.model(x, lungs)

    def training_step(self, batch, batch_idx):
        return self.shared_step(batch, batch_idx, TRAIN_PHASE)

    def validation_step(self, batch, batch_idx):
        return self.shared_step(batch, batch_idx, VALID_PHASE)

    def test_step(self, batch, batch_idx):
        return self.shared_step(batch, batch_idx, TEST_PHASE)

    # ---- epoch end (models.py:287-317 / :603-633, :367-379) ------------------------------------------------
    def training_epoch_end(self, step_outputs):
        return self.shared_epoch_end(step_outputs, TRAIN_PHASE)

    def validation_epoch_end(self, step_outputs):
        return self.shared_epoch_end(step_outputs, VALID_PHASE)

    def test_epoch_end(self, step_outputs):
        return self.shared_epoch_end(step_outputs, TEST_PHASE)

    def shared_epoch_end(self, step_outputs, phase):
        """Concatenate the step outputs, all-gather them over the ranks (utils.cat_all_gather), take the
        accuracies over everything gathered (models.py:300-301: BEFORE de-duplication), drop the samples the
        distributed sampler repeated (first occurrence per index, :303-309) and, in the train phase, rescale the
        per-class loss weights by (1 - per-class accuracy) (:367-379).  Plots / csv dumps are out of scope;
        returns what they would have been fed."""
        from .utils import cat_all_gather
        with torch.no_grad():
            cols = {k: cat_all_gather(torch.cat([o[k] for o in step_outputs]))
                    for k in ("pred_cle_labels", "cle_labels", "pred_pse_labels", "pse_labels")}
            indices = cat_all_gather(torch.cat([o["index"] for o in step_outputs]))
            acc_cle = (cols["pred_cle_labels"] == cols["cle_labels"]).float().mean()
            acc_pse = (cols["pred_pse_labels"] == cols["pse_labels"]).float().mean()
            order = torch.argsort(indices, stable=True)          # np.unique(indices, return_index=True)
            s = indices[order]
            first = torch.ones_like(s, dtype=torch.bool)
            first[1:] = s[1:] != s[:-1]
            keep = order[first]
            cols = {k: v[keep] for k, v in cols.items()}
            if phase == TRAIN_PHASE:
                for name in ("cle", "pse"):
                    w = getattr(self, f"{name}_class_weights")
                    seen = torch.unique(torch.cat([cols[f"{name}_labels"], cols[f"pred_{name}_labels"]])).numel()
                    if seen == w.numel():
                        setattr(self, f"{name}_class_weights",
                                update_class_weights(w, cols[f"{name}_labels"], cols[f"pred_{name}_labels"]))
                    else:       # the reference's element-wise product raises here (tiny epochs); keep the weights
                        import logging
                        logging.warning(f"{name}: only {seen} of {w.numel()} classes occurred this epoch; class weights kept")
            self.log(f"epoch_{phase}_acc_cle", acc_cle, on_step=False, on_epoch=True)
            self.log(f"epoch_{phase}_acc_pse", acc_pse, on_step=False, on_epoch=True)
            return dict(indices=indices[keep], acc_cle=acc_cle, acc_pse=acc_pse, **cols)

    def configure_optimizers(self):
        """models.py:381-394 / :685-698: Adam(lr=args.lr) + ExponentialLR(gamma=0.95)."""
        optimizer = FusedAdam(self.parameters(), lr=self.args.lr)
        scheduler = torch.optim.lr_scheduler.ExponentialLR(optimizer, gamma=0.95, last_epoch=-1)
        return [optimizer], [scheduler]


class ScanCLSLightningModule(_ScanModule):
    """reference models.py:160-394 (train/val step part)."""

    def shared_step(self, batch, batch_idx, stage):
        with torch.set_grad_enabled(stage == TRAIN_PHASE):
            scans = batch["image"].unsqueeze(1)
            lungs = batch["lung_mask"].unsqueeze(1).float()
            cle_labels, pse_labels = batch["cls_label"], batch["pse_label"]
            indices = batch["index"].squeeze(-1) if "index" in batch else None
            dense_outs, cls_outs = self.forward(scans, lungs)
            out = {"pred_cle_labels": cls_outs[0].detach().argmax(-1), "pred_pse_labels": cls_outs[1].detach().argmax(-1),
                   "cle_labels": cle_labels.detach(), "pse_labels": pse_labels.detach(), "index": indices}
            if stage == TRAIN_PHASE:
                dev = scans.device
                loss, parts = cls_train_loss(cls_outs, cle_labels, pse_labels, self.cle_class_weights.to(dev),
                                             self.pse_class_weights.to(dev))
                for k, v in parts.items():
                    self.log(f"{TRAIN_PHASE}_{k}", v, on_step=True, on_epoch=True, prog_bar=True)
                self.log(f"{TRAIN_PHASE}_loss", loss, on_step=True, on_epoch=True, prog_bar=True)
                out["loss"] = loss
            return out


class ScanRegLightningModule(_ScanModule):
    """reference models.py:397-698 (train/val/predict step part)."""

    def __init__(self, args):
        super().__init__(args)
        self.beta, self.gamma = BETA, GAMMA

    def shared_step(self, batch, batch_idx, stage):
        with torch.set_grad_enabled(stage == TRAIN_PHASE):
            scans = batch["image"].unsqueeze(1)
            lungs = batch["lung_mask"].unsqueeze(1).float()
            ems = batch["em_mask"].unsqueeze(1).float()
            cle_labels, pse_labels = batch["cls_label"], batch["pse_label"]
            indices = batch["index"].squeeze(-1) if "index" in batch else None
            dense_outs, reg_outs = self.forward(scans, lungs)
            out = {"pred_cle_labels": ratio_to_label(reg_outs[0].detach(), "cle"),
                   "pred_pse_labels": ratio_to_label(reg_outs[1].detach(), "pse"),
                   "cle_labels": cle_labels.detach(), "pse_labels": pse_labels.detach(), "index": indices}
            if stage == TRAIN_PHASE:
                dev = scans.device
                cw = self.cle_class_weights.to(dev)[cle_labels.long()]   # per-sample weights, models.py:556-561
                pw = self.pse_class_weights.to(dev)[pse_labels.long()]
                loss, parts = reg_train_loss(dense_outs, reg_outs, lungs, ems, cle_labels, pse_labels, cw, pw)
                for k, v in parts.items():
                    self.log(f"{TRAIN_PHASE}_{k}", v, on_step=True, on_epoch=True, prog_bar=True)
                self.log(f"{TRAIN_PHASE}_loss", loss, on_step=True, on_epoch=True, prog_bar=True)
                out["loss"] = loss
            return out

    def predict_step(self, batch, batch_idx: int, dataloader_idx: int = 0):
        """models.py:430-450: eval forward, dRAM up-projection to the scan grid x ess mask,
        percentages normalised by lungs.sum() over the WHOLE batch (:440-441)."""
        with torch.no_grad():
            scans = batch["image"].unsqueeze(1)
            lungs = batch["lung_mask"].unsqueeze(1).float()
            ess = batch["ess_mask"].unsqueeze(1).float()
            dense_outs, _ = self.forward(scans, lungs)
            B = scans.shape[0]
            size = tuple(scans.shape[-3:])
            e4 = ess.reshape(B, *size).contiguous()
            res = {}
            lung_sum = lungs.sum()
            for name, d in (("cle", dense_outs[0]), ("pse", dense_outs[1])):
                up, part = ops.upproject(d.reshape(B, *d.shape[-3:]).contiguous(), e4, size)
                res[f"{name}_dense_outs"] = up.unsqueeze(1)
                res[f"{name}_precentages"] = part.sum(1) / lung_sum
            res.update(crop_slices=batch.get("crop_slice"), original_size=batch.get("original_size"),
                       uids=batch.get("uid"))
            return res


def make_args(model_arch: str, lr: float = 1e-4, **kw) -> SimpleNamespace:
    """argparse-Namespace stand-in with the reference's flag names (train.py:20-54)."""
    return SimpleNamespace(model_arch=model_arch, lr=lr, **kw)
