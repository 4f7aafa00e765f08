"""Host helpers mirroring the hot-path parts of the reference's utils.py.

``get_model_by_name`` (reference utils.py:83-85) resolves ``conf/<name>.yaml`` ->
``_target_: med3d.<factory>`` exactly like ``hydra.utils.instantiate`` does, but without
requiring hydra/omegaconf (absent from this image): the yaml is read with PyYAML and the
``med3d.*`` target is bound to THIS package's drop-in ``med3d`` module.  If hydra is
installed and this directory is on ``sys.path`` (so that ``import med3d`` finds the
drop-in), the reference's own ``get_model_by_name`` works unchanged.
``load_state_dict_greedy`` follows reference utils.py:226-249.
"""
from __future__ import annotations

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
    """Shape-checked, key-by-key partial load (strict=False), reference utils.py:226-249."""
    model_state_dict = model.state_dict()
    for key, weight in state_dict_to_load.items():
        if key in model_state_dict:
            if model_state_dict[key].shape == weight.shape:
                logger.info(f"[load_state_dict_greedy]:correctly loading:{key}")
                model_state_dict[key] = weight
            else:
                logger.warning(f"[load_state_dict_greedy]:shape mismatch:{key}")
        else:
            logger.warning(f"[load_state_dict_greedy]:unexpected entry:{key}")
    for key in model_state_dict.keys():
        if key not in state_dict_to_load.keys():
            logger.warning(f"[load_state_dict_greedy]:missing entry:{key}")
    model.load_state_dict(model_state_dict, strict=False)
