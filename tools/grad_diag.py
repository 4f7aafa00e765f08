"""Diagnostic: per-tensor gradient error of (CPU fp32 oracle) and (HIP path) against the
fp64 oracle, to separate rounding-chaos from kernel bugs.  Usage:
   python tools/grad_diag.py resnet18segreg 1 16 32 32 [model_seed in_seed]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import golden_loss, head_weights, make_inputs
from oracle import med3d_oracle as orc
from bodyct_dram_emph_subtype_amd import med3d

factory = sys.argv[1]
shape = (int(sys.argv[2]), 1, int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]))
ms, ins = (int(sys.argv[6]), int(sys.argv[7])) if len(sys.argv) > 7 else (0, 100)
torch.manual_seed(ms)
kw = dict(n_classes=[6, 3]) if factory.endswith("cls") else {}
m = getattr(med3d, factory)(**kw)
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
names = [n for n, _ in m.named_parameters()]
x, lungs = make_inputs(ins, shape, True)
hw = head_weights(ins, shape[0])

def run_oracle(dtype):
    leaves = {k: (v.to(dtype).requires_grad_(True) if k in names else (v.to(dtype) if v.is_floating_point() else v))
              for k, v in sd0.items()}
    taps = {}
    d, o = orc.forward(leaves, x.to(dtype), lungs.to(dtype), factory, train=True, taps=taps)
    golden_loss(factory, d, o, [t.to(dtype) for t in hw]).backward()
    return {n: leaves[n].grad.double() for n in names}, [t.detach().double() for t in o], taps

g64, o64, t64 = run_oracle(torch.float64)
g32, o32, t32 = run_oracle(torch.float32)
md = m.to("cuda:0").train()
dd, od = md(x.cuda(), lungs.cuda())
golden_loss(factory, dd, od, [t.cuda() for t in hw]).backward()
gd = {n: p.grad.double().cpu() for n, p in md.named_parameters()}
rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
print("outs  cpu32-vs-64:", [rel(a, b) for a, b in zip(o32, o64)], " hip-vs-64:", [rel(a.double().cpu(), b) for a, b in zip(od, o64)])
print(f"{'param':38s} {'|g|':>10s} {'cpu32/64':>10s} {'hip/64':>10s} {'hip/cpu32':>10s}")
for n in names:
    if g64[n].norm() < 1e-6:
        continue
    print(f"{n:38s} {float(g64[n].norm()):10.3e} {rel(g32[n], g64[n]):10.2e} {rel(gd[n], g64[n]):10.2e} {rel(gd[n], g32[n]):10.2e}")
