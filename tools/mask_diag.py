"""Diagnostic: count ReLU-mask disagreements between the HIP path and the fp64 oracle per layer."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_inputs
from oracle import med3d_oracle as orc
from bodyct_dram_emph_subtype_amd import med3d

factory = sys.argv[1]
shape = (int(sys.argv[2]), 1, int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
ms, ins = int(sys.argv[6]), int(sys.argv[7])
torch.manual_seed(ms)
kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
m = getattr(med3d, factory)(**kw)
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
x, lungs = make_inputs(ins, shape, True)
taps64, taps32 = {}, {}
orc.forward({k: (v.double() if v.is_floating_point() else v) for k, v in sd0.items()}, x.double(), lungs.double(), factory, True, None, taps64)
orc.forward(sd0, x, lungs, factory, True, None, taps32)
md = m.to("cuda:0").train()
P = md._tensor_dict()
dense, outs, saved = md._engine.forward(P, x.cuda(), lungs.cuda(), True, True, None)
hip = {"stem": saved["xs"], "xup3": saved["xup3"], "xup2": saved["cu3"]["x"], "xup1": saved["cu2"][0]["x"][..., :64] * 0}
for name in ("stem", "xup2", "xup3"):
    h = hip[name].permute(0, 4, 1, 2, 3).cpu().double()
    r64, r32 = taps64[name], taps32[name].double()
    mm_h = int(((h > 0) != (r64 > 0)).sum()); mm_c = int(((r32 > 0) != (r64 > 0)).sum())
    print(f"{name:6s} n={h.numel():8d} rel hip/64 {float((h-r64).norm()/r64.norm()):.2e} cpu32/64 {float((r32-r64).norm()/r64.norm()):.2e}"
          f"  mask mismatches hip:{mm_h} cpu32:{mm_c}")
    if mm_h:
        idx = ((h > 0) != (r64 > 0)).nonzero()[:5]
        for i in idx:
            i = tuple(int(v) for v in i)
            print("     at", i, "hip", float(h[i]), "fp64", float(r64[i]), "cpu32", float(r32[i]))
# pre-BN us3 conv output statistics
y = saved["cu3"]["y"].double().cpu()
print("us3 y: mean/std per channel ratio max", float((y.mean((0,1,2,3)).abs() / y.std((0,1,2,3))).max()))
