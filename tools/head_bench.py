"""GPU box: time of the head kernels at config 1 / 2 shape (HIP events).   python tools/head_bench.py"""
import os, sys
os.environ.setdefault("DRAM_TUNING", "1")   # tuning tool: the A/B switches below count
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyct_dram_emph_subtype_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=10):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n


with ops.launch_scope(DEV):
    B, D, H, W = 2, 64, 128, 128
    for dt in (torch.float32, torch.bfloat16):
        x = torch.randn(B, D, H, W, 32, device=DEV).to(dt)
        es = x.element_size()
        for NO, sig in ((9, False), (2, True)):
            w, bias = torch.randn(NO, 32, device=DEV), torch.randn(NO, device=DEV)
            lungs = (torch.rand(B, 2 * D, 2 * H, 2 * W, device=DEV) > 0.5).float() if sig else None
            dense, _ = ops.head_fwd(x, w, bias, lungs, sig)
            gd, gp = torch.randn_like(dense), torch.randn(B, NO, device=DEV)
            n = B * D * H * W
            t = timeit(lambda: ops.head_fwd(x, w, bias, lungs, sig))
            print(f"{str(dt)[6:]:9s} NO={NO} head_fwd {t*1e3:7.1f} us  {n * (32 * es + 4 * NO) / t / 1e6:6.0f} GB/s")
            t = timeit(lambda: ops.head_bwd(x, w, dense if sig else None, gd, gp, lungs, sig))
            print(f"{str(dt)[6:]:9s} NO={NO} head_bwd {t*1e3:7.1f} us  {n * (64 * es + 4 * NO * (2 if sig else 1)) / t / 1e6:6.0f} GB/s")
