"""bodyct-dram-emph-subtype_amd -- MI355X-native hot path of DIAGNijmegen/bodyct-dram-emph-subtype.

Only what the Med3D-ResNet + dRAM train/predict step needs:
  csrc/          hand-written HIP kernels (gfx950) + the C ABI (include/dram_hip.h)
  _lib / ops     ctypes binding and tensor-level wrappers
  engine         forward/backward executor
  med3d          drop-in network factories (reference med3d.py surface)
  metrics, models, optim, distributed, utils   host-side mirrors of the reference interface

The directory name is not a Python identifier; import it as ``bodyct_dram_emph_subtype_amd``
(shim module at the repository root) or put this directory on ``sys.path`` to get drop-in
``med3d`` / ``metrics`` / ``utils`` modules for Hydra ``_target_`` strings.
"""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (does not load the .so; ops do, lazily, and fail loudly)
from .med3d import (resnet18segcls, resnet18segreg, resnet34segcls, resnet34segreg,  # noqa: F401
                    resnet50segcls, resnet50segreg)


def load_library():
    """Load libdram_hip.so now (raises if it is missing or does not export the ABI)."""
    return _lib.load()
