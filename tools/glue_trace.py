"""Where do the torch-level tiny kernels of a train step come from?  torch.profiler with stacks over one eager step:
per aten op (fill_, copy_, zero_, ...) the Python call sites.   python tools/glue_trace.py [config]"""
import collections, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import bodyct_dram_emph_subtype_amd as dram
from bodyct_dram_emph_subtype_amd import med3d
from bodyct_dram_emph_subtype_amd.optim import FusedAdam
from torch.profiler import profile, ProfilerActivity

cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dram.load_library()
factory, B, dims, *_ = bench.CONFIGS[cfg]
dev = torch.device("cuda", 0)
torch.manual_seed(0)
kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
m = getattr(med3d, factory)(**kw).to(dev).train()
if cfg in bench.BF16_CONFIGS:
    m.storage_dtype = torch.bfloat16
opt = FusedAdam(m.parameters(), lr=1e-4)
batch = bench.synth_batch(B, dims, 0, dev)
step = bench.make_step(factory, m, opt, batch)
for _ in range(3):
    step()
torch.cuda.synchronize()
import traceback
from torch.utils._python_dispatch import TorchDispatchMode

sites = collections.defaultdict(collections.Counter)


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("_to_copy", "copy_", "fill_", "zero_", "zeros", "clone", "cat", "stack", "mul", "add", "div", "sum")):
            st = [f"{os.path.basename(f.filename)}:{f.lineno} {f.name}" for f in traceback.extract_stack()
                  if ("bodyct" in f.filename or "bench.py" in f.filename) and "glue_trace" not in f.filename]
            sites[name][" <- ".join(reversed(st[-3:])) if st else "(autograd engine thread / C++)"] += 1
        return func(*args, **(kwargs or {}))


with Rec():
    step()
    torch.cuda.synchronize()
for name, c in sorted(sites.items(), key=lambda kv: -sum(kv[1].values())):
    print(name, sum(c.values()))
    for s_, n in c.most_common(10):
        print(f"   {n:4d}  {s_[:200]}")
