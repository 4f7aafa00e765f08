import os
import sys

import pytest
import torch

# the tests force kernel variants (DRAM_CONV_ALGO, DRAM_W2D_V, DRAM_WINO_TILING, ...); the library and the host code
# read those switches only under DRAM_TUNING=1 (tests/test_host.py checks that they are ignored without it)
os.environ["DRAM_TUNING"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def _host_threads():
    """The GPU box exposes every hardware thread of the host (torch defaults to 128 of them) but the job owns a
    16-CPU share (cgroup cpu.max): the fp64 oracle convolutions run ~25 % faster on 2 x the share than
    oversubscribed (tools/cpu_threads_probe.py)."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            return max(1, min(os.cpu_count() or 1, 2 * int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return None


_n = _host_threads()
if _n:
    torch.set_num_threads(_n)


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
