"""GPU-side deterministic input pipeline of the reference's ``SubtypeDataModule._transform``
(reference models.py:55-63): IntensityWindow(from_span=(-1150,-300), to_span=(0,1)) ->
Standardize() -> Interpolate(target_size, align_corners=True, only_in_plane=True); masks take the
nearest-neighbour branch with the same depth indices.  Fused HIP kernels (csrc/prep.hip): one
reduction pass for the volume statistics, one fused window/standardize/resize pass.

The train-time random augmentations of models.py:66-74 (GaussianAddictive, BoxMaskOut, Flip, CropAndResize)
run as ONE fused gather kernel per volume (`dram_augment_image` / `dram_augment_mask`): ``TrainAugment`` draws
the parameters on the host with the reference's distributions (its ``get_params``), the kernels apply them;
with given parameters they reproduce the reference classes (tests/golden/augment.npz).
"""
from __future__ import annotations

if not __package__:          # imported top-level (this directory on sys.path): bind to the package, see _dropin.py
    import _dropin
    __package__ = _dropin.adopt(__name__)

import ctypes
import random
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

import torch

from . import ops
from .ops import _L, _chk, _p, _req, _stream

FROM_SPAN = (-1150.0, -300.0)      # models.py:60


def depth_indices(D: int, new_d: int, device) -> torch.Tensor:
    """spatial_transforms.py:66: torch.linspace(0, original_d - 1, new_d).long()"""
    return torch.linspace(0, D - 1, new_d).long().to(torch.int32).to(device)


def prepare_image(scan: torch.Tensor, target_size: Sequence[int], from_span=FROM_SPAN) -> torch.Tensor:
    """scan: raw HU volume [D,H,W] (any real dtype, device tensor) -> float32 [Do,Ho,Wo]."""
    scan = scan.float().contiguous()
    _req(scan, "scan")
    D, H, W = scan.shape
    Do, Ho, Wo = (int(v) for v in target_size)
    n = scan.numel()
    nblk = _L().dram_window_stats_nblk(n)
    partial = torch.empty((nblk, 2), device=scan.device, dtype=torch.float32)
    lo, hi = float(from_span[0]), float(from_span[1])
    _chk(_L().dram_window_stats(_p(scan), _p(partial), n, lo, hi, _stream()), "dram_window_stats")
    s = partial.double().sum(0)                                   # O(1) glue on [nblk,2]
    mean = s[0] / n
    var = (s[1] - n * mean * mean) / (n - 1)                       # torch.std(): unbiased
    mean_inv = torch.stack([mean, var.clamp_min(0).rsqrt()]).float().contiguous()
    zidx = depth_indices(D, Do, scan.device)
    out = torch.empty((Do, Ho, Wo), device=scan.device, dtype=torch.float32)
    _chk(_L().dram_prep_image(_p(scan), _p(zidx), _p(mean_inv), _p(out), D, H, W, Do, Ho, Wo, lo, hi, _stream()),
         "dram_prep_image")
    return out


def prepare_mask(mask: torch.Tensor, target_size: Sequence[int]) -> torch.Tensor:
    """mask [D,H,W] (bool / int / float) -> same dtype [Do,Ho,Wo], nearest in-plane + depth select."""
    dtype = mask.dtype
    m = mask.float().contiguous()
    _req(m, "mask")
    D, H, W = m.shape
    Do, Ho, Wo = (int(v) for v in target_size)
    zidx = depth_indices(D, Do, m.device)
    out = torch.empty((Do, Ho, Wo), device=m.device, dtype=torch.float32)
    _chk(_L().dram_prep_mask(_p(m), _p(zidx), _p(out), D, H, W, Do, Ho, Wo, _stream()), "dram_prep_mask")
    return out.to(dtype)


def prepare_sample(sample: Dict[str, torch.Tensor], target_size: Sequence[int]) -> Dict[str, torch.Tensor]:
    """Dict transform with the reference's keys: 'image' + '*_mask' entries (base.py dict transforms)."""
    out = dict(sample)
    for k, v in sample.items():
        if k == "image":
            out[k] = prepare_image(v, target_size)
        elif k.endswith("_mask"):
            out[k] = prepare_mask(v, target_size)
    return out


# --------------------------------------------------------------------------- train-time augmentations
def _frac_box(center, size, shape):
    """integer box of BoxMaskOut / CropAndResize (intensity_transforms.py:226-235, spatial_transforms.py:172-177)"""
    return [(max(0, int(mc * ds) - int(ms * ds) // 2), min(int(mc * ds) + (int(ms * ds) - int(ms * ds) // 2), ds))
            for mc, ds, ms in zip(center, shape, size)]


@dataclass
class AugmentParams:
    """One draw of the four train-time transforms (None / empty = that transform is not applied)."""
    noise_sigma: Optional[float] = None                                   # GaussianAddictive
    box_centers: List[Tuple[float, float, float]] = field(default_factory=list)   # BoxMaskOut
    box_sizes: List[Tuple[float, float, float]] = field(default_factory=list)
    flip_dims: Tuple[int, ...] = ()                                       # Flip (axes of the [D,H,W] volume)
    crop_center: Optional[Tuple[float, float, float]] = None              # CropAndResize
    crop_size: Optional[Tuple[float, float, float]] = None

    def to_struct(self, shape) -> "ops._lib.DramAugment":
        a = ops._lib.DramAugment()
        flags = 0
        if self.noise_sigma is not None:
            flags |= 1
            a.sigma = float(self.noise_sigma)
        if self.box_centers:
            if len(self.box_centers) > 10:
                raise ValueError("BoxMaskOut: at most 10 boxes (reference n_masks=(1, 10))")
            flags |= 2
            a.n_boxes = len(self.box_centers)
            for b, (c, sz) in enumerate(zip(self.box_centers, self.box_sizes)):
                (z0, z1), (y0, y1), (x0, x1) = _frac_box(c, sz, shape)
                for k, v in enumerate((z0, z1, y0, y1, x0, x1)):
                    a.boxes[b][k] = v
        if self.flip_dims:
            flags |= 4
            a.flip_axes = sum(1 << int(d) for d in set(self.flip_dims))
        if self.crop_center is not None:
            flags |= 8
            for k, ((lo, hi), n) in enumerate(zip(_frac_box(self.crop_center, self.crop_size, shape), shape)):
                # torch: as_tensor(int box) / as_tensor(size) in float32
                a.box_lo[k] = float(np.float32(lo) / np.float32(n))
                a.box_hi[k] = float(np.float32(hi) / np.float32(n))
        a.flags = flags
        return a


def augment_image(image: torch.Tensor, params: AugmentParams, noise: Optional[torch.Tensor] = None) -> torch.Tensor:
    """image [D,H,W] float32 device tensor -> augmented copy.  `noise` [D,H,W]: the N(0,1) draw of
    GaussianAddictive (drawn on the device when omitted)."""
    image = image.float().contiguous()
    _req(image, "image")
    D, H, W = image.shape
    a = params.to_struct(image.shape)
    mm = None
    if a.flags & 1:
        if noise is None:
            noise = torch.randn(image.shape, device=image.device)
        _req(noise, "noise", shape=image.shape)
        nblk = _L().dram_minmax_nblk(image.numel())
        part = torch.empty((nblk, 2), device=image.device, dtype=torch.float32)
        _chk(_L().dram_minmax(_p(image), _p(part), image.numel(), _stream()), "dram_minmax")
        mm = torch.stack([part[:, 0].amin(), part[:, 1].amax()]).contiguous()     # O(nblk) glue
    out = torch.empty_like(image)
    _chk(_L().dram_augment_image(_p(image), _p(noise), _p(mm), _p(out), D, H, W, ctypes.byref(a), _stream()),
         "dram_augment_image")
    return out


def augment_mask(mask: torch.Tensor, params: AugmentParams) -> torch.Tensor:
    """mask [D,H,W] -> Flip + CropAndResize(nearest) with the same parameters (DualTransform semantics)."""
    dtype = mask.dtype
    m = mask.float().contiguous()
    _req(m, "mask")
    D, H, W = m.shape
    a = params.to_struct(m.shape)
    if not (a.flags & 12):
        return mask
    out = torch.empty_like(m)
    _chk(_L().dram_augment_mask(_p(m), _p(out), D, H, W, ctypes.byref(a), _stream()), "dram_augment_mask")
    return out.to(dtype)


class TrainAugment:
    """The augmentation list of models.py:66-74 with the reference's probabilities and parameter ranges:
    GaussianAddictive(p=.5, sigma (0.03, 0.06)), BoxMaskOut(p=.5, 1-10 boxes, centres (0.2, 0.8), sizes
    (0.01, 0.06)), Flip(p=.5, 1-2 of the 3 axes), CropAndResize(p=.5, centre (0.45, 0.55), size (0.95, 1.0)).
    Parameters are drawn on the host like the reference's get_params (same distributions; the streams of
    Python's / numpy's global generators are not reproduced)."""

    def __init__(self, p: float = 0.5, rng: Optional[random.Random] = None):
        self.p = p
        self.rng = rng or random.Random()

    def draw(self) -> AugmentParams:
        r, ap = self.rng, AugmentParams()
        if r.random() < self.p:
            ap.noise_sigma = r.uniform(0.03, 0.06)
        if r.random() < self.p:
            n = r.randint(1, 10)
            ap.box_centers = [tuple(r.uniform(0.2, 0.8) for _ in range(3)) for _ in range(n)]
            ap.box_sizes = [tuple(r.uniform(0.01, 0.06) for _ in range(3)) for _ in range(n)]
        if r.random() < self.p:
            ap.flip_dims = tuple(r.sample(range(3), r.randint(1, 2)))      # np.random.randint(1, 3) axes
        if r.random() < self.p:
            ap.crop_center = tuple(r.uniform(0.45, 0.55) for _ in range(3))
            ap.crop_size = tuple(r.uniform(0.95, 1.0) for _ in range(3))
        return ap

    def __call__(self, sample: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        ap = self.draw()
        out = dict(sample)
        for k, v in sample.items():
            if k == "image":
                out[k] = augment_image(v, ap)
            elif k.endswith("_mask"):
                out[k] = augment_mask(v, ap)
        return out
