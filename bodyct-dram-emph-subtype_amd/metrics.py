"""Drop-in for the reference's metrics.py (BinaryDice / BinaryCrossEntropy / dice_coef), backed by
the fused HIP loss kernels (csrc/head_loss.hip).

Same call signatures as the reference: ``BinaryCrossEntropy()(y, y_hat, mask=None, smoothness=0.65)``
(metrics.py:10), ``BinaryDice(smooth)(y, y_hat, extra_args={})`` (metrics.py:40-47) and
``dice_coef(y, y_hat, smooth)`` (metrics.py:33-37), on device tensors of any (equal) shape, differentiable
w.r.t. the prediction(s).  The train step does not call these one at a time -- it uses
``models.segmentation_loss`` (ONE fused pass for both terms, with the ``clamp(cle + pse)`` of models.py:527
and the nearest-resize of the masks folded in); each class here runs that same kernel pair with the
other term's operands neutralised, so code written against the reference's ``metrics`` keeps working.
"""
from __future__ import annotations

if not __package__:          # imported top-level (this directory on sys.path): bind to the package, see _dropin.py
    import _dropin
    __package__ = _dropin.adopt(__name__)

import torch

from .models import _SegLossFn


def _flat(t: torch.Tensor, batch: bool) -> torch.Tensor:
    """[B, ...] -> [B,1,1,n] (or [1,1,1,N]): the kernels take 4-D maps; the loss sums are shape-agnostic."""
    t = t.float()
    return t.reshape(t.shape[0] if batch else 1, 1, 1, -1).contiguous()


def dice_coef(y, y_hat, smooth):
    """reference metrics.py:33-37: (2 sum(y*y_hat) + smooth) / (sum y + sum y_hat + smooth)."""
    if y.numel() != y_hat.numel():
        raise ValueError(f"dice_coef: {tuple(y.shape)} vs {tuple(y_hat.shape)}")
    a, b = _flat(y, False), _flat(y_hat, False)
    ones = torch.ones_like(a)
    mul, _ = _SegLossFn.apply(a, b, ones, torch.zeros_like(a), torch.zeros(1, device=a.device), float(smooth), 0.85)
    return mul


class BinaryDice:
    """reference metrics.py:40-47."""

    def __init__(self, smooth):
        self.smooth = smooth

    def __call__(self, y, y_hat, extra_args={}):
        return dice_coef(y, y_hat, self.smooth)


class BinaryCrossEntropy:
    """reference metrics.py:4-30: class-balanced (alpha = clamp(1 - sum(t)/N, 0.3, 0.7)) BCE of the
    probabilities y_hat against the targets y, weighted ``smoothness`` inside ``mask`` and 1 outside it
    (everywhere ``smoothness`` when mask is None), normalised by the sum of the class weights."""

    def __init__(self):
        self.eps = 1e-6

    def __call__(self, y, y_hat, mask=None, smoothness=0.65):
        assert y.size() == y_hat.size()
        t, p = _flat(y, True), _flat(y_hat, True)
        m = torch.ones_like(t) if mask is None else _flat(mask.expand_as(y), True)
        ones = torch.ones(t.shape[0], device=t.device)
        # the kernel's clamp(cle + pse, 0, 1) with pse = 0: for targets in {0, 1} a no-op in value and gradient,
        # because metrics.py:22 clamps pt = p or 1 - p to [eps, 1 - eps] anyway
        _, bce = _SegLossFn.apply(p, torch.zeros_like(p), m, t, ones, 1e-7, float(smoothness))
        return bce
