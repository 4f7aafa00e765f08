"""GPU-side deterministic input pipeline of the reference's ``SubtypeDataModule._transform``
(reference models.py:55-63): IntensityWindow(from_span=(-1150,-300), to_span=(0,1)) ->
Standardize() -> Interpolate(target_size, align_corners=True, only_in_plane=True); masks take the
nearest-neighbour branch with the same depth indices.  Fused HIP kernels (csrc/prep.hip): one
reduction pass for the volume statistics, one fused window/standardize/resize pass.

The train-time random augmentations (GaussianAddictive, BoxMaskOut, Flip, CropAndResize,
models.py:66-74) stay out of scope.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Sequence

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
