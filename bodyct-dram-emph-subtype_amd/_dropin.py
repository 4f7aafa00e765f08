"""Top-level import support for the reference's module names.

The reference resolves its network through Hydra: ``conf/med3ddram18.yaml`` says
``_target_: med3d.resnet18segreg`` and ``hydra.utils.instantiate`` does
``importlib.import_module("med3d")`` (reference utils.py:83-85); its own files say ``from metrics
import ...``, ``from utils import ...``, ``from models import ...``.  With THIS directory on
``sys.path`` (``PYTHONPATH=<repo>/bodyct-dram-emph-subtype_amd``) those imports find the drop-in files
here, executed as top-level modules -- where package-relative imports would fail.  Each drop-in
module therefore starts with

    if not __package__:
        import _dropin
        __package__ = _dropin.adopt(__name__)

``adopt`` imports the real package (through the identifier shim at the repository root), makes
``sys.modules[<name>]`` the package's module of that name -- so ``import med3d`` and
``bodyct_dram_emph_subtype_amd.med3d`` are ONE module object with one set of classes -- and
returns the package name, which lets the rest of the file's relative imports resolve.
"""
import importlib
import os
import sys

PACKAGE = "bodyct_dram_emph_subtype_amd"


def adopt(name: str) -> str:
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if root not in sys.path:
        sys.path.append(root)
    importlib.import_module(PACKAGE)
    sys.modules[name] = importlib.import_module(f"{PACKAGE}.{name}")
    return PACKAGE
