"""Drop-in for the reference's metrics.py (BinaryDice / BinaryCrossEntropy), backed by the
fused HIP loss kernels.  The train step does not call these one at a time -- it uses
``models.segmentation_loss`` (one fused pass for both terms); they exist so that code
written against ``metrics.BinaryDice(1e-7)`` / ``metrics.BinaryCrossEntropy()`` keeps
working on device tensors, and compute exactly the two halves of that fused pass.
"""
from __future__ import annotations

import torch

from .models import _SegLossFn


def _as4(t):
    return t.reshape(t.shape[0], *t.shape[-3:]).contiguous().float()


class BinaryDice:
    """reference metrics.py:40-47; dice_coef(y, y_hat, smooth) of two [B,1,D,H,W] maps."""

    def __init__(self, smooth):
        if abs(smooth - 1e-7) > 1e-12:
            raise NotImplementedError("the fused kernel implements smooth=1e-7 (models.py:412)")
        self.smooth = smooth

    def __call__(self, y, y_hat, extra_args={}):
        ones = torch.ones_like(_as4(y))
        zeros = torch.zeros(y.shape[0], device=y.device)
        mul, _ = _SegLossFn.apply(_as4(y), _as4(y_hat), ones, torch.zeros_like(ones), zeros)
        return mul


class BinaryCrossEntropy:
    """reference metrics.py:4-30 with smoothness=0.85 and a lung mask (the only call site,
    models.py:529): y = target, y_hat = clamp(cle+pse, 0, 1) given as the SUM of two maps."""

    def __init__(self):
        self.eps = 1e-6

    def __call__(self, y, cle, pse, mask, smoothness=0.85):
        if abs(smoothness - 0.85) > 1e-12:
            raise NotImplementedError("the fused kernel implements smoothness=0.85 (models.py:529)")
        ones = torch.ones(y.shape[0], device=y.device)
        _, seg = _SegLossFn.apply(_as4(cle), _as4(pse), _as4(mask), _as4(y), ones)
        return seg
