"""GPU box: achieved HBM GB/s of the element-wise kernels at the config-1 / config-2 shapes, fp32 and bf16
storage.   python tools/ew_bench.py   (DRAM_EW_U=0|1|2|4 selects the bn_apply variant)"""
import os
os.environ.setdefault("DRAM_TUNING", "1")   # tuning tool: the A/B switches below count
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bodyct_dram_emph_subtype_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, n=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    with ops.launch_scope(DEV):
        for dt in (torch.float32, torch.bfloat16):
            es = 4 if dt == torch.float32 else 2
            for shape in ((2, 64, 128, 128, 64), (2, 32, 64, 64, 64), (2, 16, 32, 32, 512)):
                n = 1
                for v in shape:
                    n *= v
                C = shape[-1]
                y = torch.randn(shape, device=DEV).to(dt)
                dz = torch.randn(shape, device=DEV).to(dt)
                res = torch.randn(shape, device=DEV).to(dt)
                sc, sh = torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
                mean, invstd, gamma = torch.randn(C, device=DEV), torch.rand(C, device=DEV) + 0.5, torch.randn(C, device=DEV)
                rows = n // C
                t = timeit(lambda: ops.bn_apply(y, sc, sh, None, 1, True))
                print(f"{str(dt)[6:]:9s} {str(shape):24s} bn_apply          {t:7.3f} ms  {2 * n * es / t / 1e6:7.0f} GB/s")
                t = timeit(lambda: ops.bn_apply(y, sc, sh, res, 1, True))
                print(f"{str(dt)[6:]:9s} {str(shape):24s} bn_apply+res      {t:7.3f} ms  {3 * n * es / t / 1e6:7.0f} GB/s")
                t = timeit(lambda: ops.bn_bwd_reduce(dz, None, y, mean, invstd, True, sc, sh))
                print(f"{str(dt)[6:]:9s} {str(shape):24s} bn_bwd_reduce     {t:7.3f} ms  {2 * n * es / t / 1e6:7.0f} GB/s")
                part = ops.bn_bwd_reduce(dz, None, y, mean, invstd, True, sc, sh)
                sums = ops.reduce_partials(part)
                t = timeit(lambda: ops.bn_bwd_apply(dz, None, y, mean, invstd, gamma, sums, float(rows), True, sc, sh))
                print(f"{str(dt)[6:]:9s} {str(shape):24s} bn_bwd_apply      {t:7.3f} ms  {3 * n * es / t / 1e6:7.0f} GB/s")
                if C == 64 and shape[1] == 64:
                    t = timeit(lambda: ops.maxpool_fwd(y))
                    print(f"{str(dt)[6:]:9s} {str(shape):24s} maxpool_fwd       {t:7.3f} ms  {(n * es * 1.125 + n / 8) / t / 1e6:7.0f} GB/s")
                    p, am = ops.maxpool_fwd(y)
                    t = timeit(lambda: ops.maxpool_bwd(p, am, tuple(y.shape), res))
                    print(f"{str(dt)[6:]:9s} {str(shape):24s} maxpool_bwd+add   {t:7.3f} ms  {(n * es * 2.125 + n / 8) / t / 1e6:7.0f} GB/s")
                if C == 64 and shape[1] == 32:
                    skip = torch.randn((2, 64, 128, 128, 64), device=DEV).to(dt)
                    t = timeit(lambda: ops.upcat_fwd(y, skip))
                    nb = (n + skip.numel() * 2 + n * 8) * es
                    print(f"{str(dt)[6:]:9s} {str(shape):24s} upcat_fwd         {t:7.3f} ms  {nb / t / 1e6:7.0f} GB/s")
                    cat = ops.upcat_fwd(y, skip)
                    t = timeit(lambda: ops.upcat_bwd(cat, tuple(y.shape), tuple(skip.shape), True, False))
                    print(f"{str(dt)[6:]:9s} {str(shape):24s} upcat_bwd_src     {t:7.3f} ms  {(n * 8 + n) * es / t / 1e6:7.0f} GB/s")
                del y, dz, res


if __name__ == "__main__":
    main()
