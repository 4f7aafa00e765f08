"""GPU box: per-launch view of the library's kernel timeline for one train step of a bench config
   python tools/timeline_dump.py [config] [family substring] [--bf16]
prints, per (family, variant, flops, bytes) group: launches/step, avg us, executed TFLOP/s, algorithmic GB/s."""
import collections
import os
os.environ.setdefault("DRAM_TUNING", "1")   # tuning tool: the A/B switches below count
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from bodyct_dram_emph_subtype_amd import _lib, med3d, ops  # noqa: E402
from bodyct_dram_emph_subtype_amd.optim import FusedAdam  # noqa: E402

BF16 = "--bf16" in sys.argv
args = [a for a in sys.argv[1:] if a != "--bf16"]
cfg = int(args[0]) if len(args) > 0 else 1
pat = args[1] if len(args) > 1 else ""
os.environ["DRAM_WGRAD_STREAM"] = "0"
factory, B, dims, *_ = bench.CONFIGS[cfg]
torch.manual_seed(0)
m = getattr(med3d, factory)(**(dict(n_classes=[6, 3]) if factory.endswith("cls") else {})).cuda().train()
if BF16:
    m.storage_dtype = torch.bfloat16
opt = FusedAdam(m.parameters(), lr=1e-4)
step = bench.make_step(factory, m, opt, bench.synth_batch(B, dims, 0, "cuda"))
for _ in range(3):
    step()
torch.cuda.synchronize()
N = 3
L = _lib.load()
L.dram_profile_start(8192 * N)
for _ in range(N):
    step()
torch.cuda.synchronize()
L.dram_profile_stop()
buf = (_lib.DramProfRecord * (8192 * N))()
n = L.dram_profile_read(buf, 8192 * N)
g = collections.OrderedDict()
for i in range(n):
    r = buf[i]
    k = (L.dram_profile_family_name(r.family).decode(), r.variant, r.mfma_flops, r.hbm_bytes)
    d = g.setdefault(k, [0, 0.0])
    d[0] += 1
    d[1] += r.ms
rows = sorted(g.items(), key=lambda kv: -kv[1][1])
print(f"{'family':16s} {'variant':>8s} {'n/step':>6s} {'avg us':>9s} {'ms/step':>8s} {'TFLOP/s':>8s} {'GB/s':>8s}")
for (fam, var, fl, by), (cnt, ms) in rows:
    if pat and pat not in fam:
        continue
    avg = ms / cnt
    print(f"{fam:16s} {var:8d} {cnt / N:6.1f} {avg * 1e3:9.1f} {ms / N:8.3f} {fl / avg / 1e9:8.1f} {by / avg / 1e6:8.0f}")
