import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size workload")


def pytest_collection_modifyitems(config, items):
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def dram():
    import bodyct_dram_emph_subtype_amd as pkg
    return pkg


def make_inputs(seed, shape, with_lungs=True):
    """Same recipe as tests/golden/make_golden.py::make_inputs."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(*shape, generator=g)
    lungs = (torch.rand(*shape, generator=g) > 0.3).float() if with_lungs else None
    return x, lungs


def head_weights(seed, B):
    g = torch.Generator().manual_seed(seed + 77)
    return [torch.randn(B, 6, generator=g), torch.randn(B, 3, generator=g), torch.randn(B, generator=g),
            torch.randn(B, generator=g)]


def golden_loss(factory, dense, outs, hw):
    """The scalar objective the golden nets were differentiated through (make_golden.py::net_case)."""
    if factory.endswith("cls"):
        return (outs[0] * hw[0]).sum() + (outs[1] * hw[1]).sum()
    return (outs[0] * hw[2]).sum() + (outs[1] * hw[3]).sum() + 0.1 * (dense[0] * dense[1]).mean()


def rel_l2(a, b):
    a = torch.as_tensor(a).double().flatten()
    b = torch.as_tensor(b).double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
