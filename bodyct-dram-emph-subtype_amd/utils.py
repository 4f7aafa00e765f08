"""Host helpers mirroring the hot-path parts of the reference's utils.py.

``get_model_by_name`` (reference utils.py:83-85) resolves ``conf/<name>.yaml`` ->
``_target_: med3d.<factory>`` exactly like ``hydra.utils.instantiate`` does, but without
requiring hydra/omegaconf (absent from this image): the yaml is read with PyYAML and the
``med3d.*`` target is bound to THIS package's drop-in ``med3d`` module.  If hydra is
installed and this directory is on ``sys.path`` (so that ``import med3d`` finds the
drop-in), the reference's own ``get_model_by_name`` works unchanged.
``load_state_dict_greedy`` keeps the semantics of reference utils.py:226-249 (own implementation).
"""
from __future__ import annotations

if not __package__:          # imported top-level (this directory on sys.path): bind to the package, see _dropin.py
    import _dropin
    __package__ = _dropin.adopt(__name__)

import importlib
import logging
import os
from typing import Dict

import torch
import yaml

logger = logging.getLogger(__name__)
_CONF_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "conf")


def get_model_by_name(name: str, conf_dir: str = None):
    # reference: OmegaConf.load(f"./conf/{name}.yaml") (cwd-relative); fall back to the
    # package's own conf/ mirror
    candidates = [os.path.join(conf_dir, f"{name}.yaml")] if conf_dir else []
    candidates += [os.path.join(".", "conf", f"{name}.yaml"), os.path.join(_CONF_DIR, f"{name}.yaml")]
    for path in candidates:
        if os.path.exists(path):
            break
    else:
        raise FileNotFoundError(f"no conf/{name}.yaml (looked in {candidates})")
    with open(path) as f:
        cfg = yaml.safe_load(f)
    target = cfg.pop("_target_")
    mod_name, fn_name = target.rsplit(".", 1)
    if mod_name == "med3d":
        from . import med3d as mod
    else:
        mod = importlib.import_module(mod_name)
    return getattr(mod, fn_name)(**cfg)


def load_state_dict_greedy(model: torch.nn.Module, state_dict_to_load: Dict):
    """Partial, shape-checked load with the semantics of reference utils.py:226-249: an entry is taken
    when the model has a tensor of that name AND shape; everything else (unknown names, shape
    mismatches, names the checkpoint lacks) is skipped and reported, never an error.  Returns the
    three skipped groups so callers / tests can inspect what happened."""
    have = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    usable = {k: v for k, v in state_dict_to_load.items() if have.get(k) == tuple(v.shape)}
    mismatched = sorted(k for k in state_dict_to_load if k in have and k not in usable)
    unexpected = sorted(k for k in state_dict_to_load if k not in have)
    missing = sorted(k for k in have if k not in state_dict_to_load)
    model.load_state_dict(usable, strict=False)
    report = logger.warning if (mismatched or unexpected or missing) else logger.info
    report("[load_state_dict_greedy] loaded %d/%d tensors; shape mismatch: %s; unexpected: %s; missing: %s",
           len(usable), len(have), mismatched or "-", unexpected or "-", missing or "-")
    return dict(mismatched=mismatched, unexpected=unexpected, missing=missing)


def cat_all_gather(t: torch.Tensor) -> torch.Tensor:
    """utils.py:66-80: all_gather + concatenate along dim 0 (identity without an initialised process group --
    the reference requires one)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return t
    parts = [torch.ones_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(parts, t, async_op=False)
    return torch.cat(parts, dim=0)
