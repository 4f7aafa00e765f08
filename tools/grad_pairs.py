"""GPU box: pairwise gradient distances between the HIP path and four CPU oracle evaluations
(fp32 / fp64, free / pinned to the HIP forward's ReLU + max-pool decisions) for one train step.
   python tools/grad_pairs.py [factory] [D H W] [loss: dram|golden] [head bias, e.g. -1]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import rel_l2  # noqa: E402
from oracle import med3d_oracle as orc  # noqa: E402
from bodyct_dram_emph_subtype_amd import med3d, models  # noqa: E402
from bodyct_dram_emph_subtype_amd.engine import forward_decisions  # noqa: E402

factory = sys.argv[1] if len(sys.argv) > 1 else "resnet18segreg"
dims = tuple(int(v) for v in sys.argv[2:5]) if len(sys.argv) > 4 else (64, 128, 128)
loss_kind = sys.argv[5] if len(sys.argv) > 5 else "dram"
torch.manual_seed(5)
m = getattr(med3d, factory)(**(dict(n_classes=[6, 3]) if factory.endswith("cls") else {}))
ONLY_PINNED = os.environ.get("GP_ONLY_PINNED", "0") == "1"
if len(sys.argv) > 6:     # move cle+pse away from the clamp(., 0, 1) kink of models.py:527 (sigmoid(0)*2 == 1 at init)
    with torch.no_grad():
        for fc in m.fcs:
            fc.bias.fill_(float(sys.argv[6]))
sd0 = {k: v.clone() for k, v in m.state_dict().items()}
names = [n for n, _ in m.named_parameters()]
g = torch.Generator().manual_seed(11)
x = torch.randn(1, 1, *dims, generator=g)
D, H, W = dims
z = (torch.arange(D).float() - (D - 1) / 2) / (0.4 * D)
y = (torch.arange(H).float() - (H - 1) / 2) / (0.35 * H)
xx = (torch.arange(W).float() - (W - 1) / 2) / (0.4 * W)
lungs = ((z[:, None, None] ** 2 + y[None, :, None] ** 2 + xx[None, None, :] ** 2) <= 1.0).float()[None, None].contiguous()
ems = ((x < -1.0).float() * lungs)
cle, pse = torch.tensor([3]), torch.tensor([1])
cw, pw = torch.tensor([0.3]), torch.tensor([0.6])


def the_loss(mod, d, o, dev, dt):
    t = lambda v: v.to(dev)
    if loss_kind == "golden":
        return o[0].sum() * 0.7 - o[1].sum() * 1.3 + 0.1 * (d[0] * d[1]).mean()
    if loss_kind == "ce":
        return mod.cls_train_loss(o, t(torch.tensor([4])), t(torch.tensor([0])), t(torch.full((6,), 1 / 6)).to(dt),
                                  t(torch.full((3,), 1 / 3)).to(dt))[0]
    return mod.reg_train_loss(d, o, t(lungs).to(dt), t(ems).to(dt), t(cle), t(pse), t(cw).to(dt), t(pw).to(dt))[0]


def oracle(dtype, pins=None):
    lv = {k: (v.clone().to(dtype).requires_grad_(True) if k in names
              else (v.clone().to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
    d, o = orc.forward(lv, x.to(dtype), lungs.to(dtype), factory, train=True, pins=pins)
    loss = the_loss(orc, d, o, "cpu", dtype)
    loss.backward()
    return {n: lv[n].grad.double() for n in names}, float(loss)


md = m.to("cuda:0").train()
dd, od = md(x.cuda(), lungs.cuda())
pins = {k: v.cpu() for k, v in forward_decisions(dd[0].grad_fn.saved_state).items()}
loss = the_loss(models, dd, od, "cuda:0", torch.float32)
loss.backward()
G = {"hip": ({n: p.grad.double().cpu() for n, p in md.named_parameters()}, float(loss))}
if not ONLY_PINNED:
    G["f32"] = oracle(torch.float32)
    G["f64"] = oracle(torch.float64)
G["f32pin"] = oracle(torch.float32, pins)
G["f64pin"] = oracle(torch.float64, pins)
print("loss:", {k: v[1] for k, v in G.items()})
sel = ["conv1.weight", "bn1.weight", "layer1.0.conv1.weight", "layer2.0.conv1.weight", "layer4.1.conv2.weight",
       "us1.conv_blocks.0.0.weight", "us1.conv_blocks.0.1.weight", "us1.conv_blocks.1.0.weight",
       "us2.conv_blocks.0.0.weight", "us2.conv_blocks.1.0.weight", "us3.0.weight", "us3.1.weight", "fcs.0.weight"]
sel = [n for n in sel if n in names]
keys = list(G)
for n in sel:
    row = "  ".join(f"{a}-{b}:{rel_l2(G[a][0][n], G[b][0][n]):.1e}" for i, a in enumerate(keys) for b in keys[i + 1:])
    print(f"{n:32s} |g|={float(G['f64pin'][0][n].norm()):.3e}  {row}")

# ---- split view of the dRAM path: (1) loss kernels' gradient FIELDS at the HIP forward's own outputs vs fp64,
#      (2) network backward of ONE shared upstream field: HIP vs pinned fp64 / pinned fp32 oracles
if loss_kind == "dram":
    md.zero_grad()
    dd, od = md(x.cuda(), lungs.cuda())
    pins = {k: v.cpu() for k, v in forward_decisions(dd[0].grad_fn.saved_state).items()}
    loss = the_loss(models, dd, od, "cuda:0", torch.float32)
    ups = torch.autograd.grad(loss, dd + od, retain_graph=True)
    leaf = [t.detach().cpu().double().requires_grad_(True) for t in dd + od]
    l64 = orc.reg_train_loss(leaf[:2], leaf[2:], lungs.double(), ems.double(), cle, pse, cw.double(), pw.double())[0]
    l64.backward()
    print("upstream fields (HIP loss kernels vs fp64 at the same dense/outs):",
          [f"{rel_l2(u.cpu(), t.grad):.1e}" for u, t in zip(ups, leaf)])
    torch.autograd.backward(dd + od, list(ups))
    hip = {n: p.grad.double().cpu() for n, p in md.named_parameters()}

    def oracle_up(dtype):
        lv = {k: (v.clone().to(dtype).requires_grad_(True) if k in names
                  else (v.clone().to(dtype) if v.is_floating_point() else v.clone())) for k, v in sd0.items()}
        d, o = orc.forward(lv, x.to(dtype), lungs.to(dtype), factory, train=True, pins=pins)
        torch.autograd.backward(d + o, [u.cpu().to(dtype) for u in ups])
        return {n: lv[n].grad.double() for n in names}
    u64, u32 = oracle_up(torch.float64), oracle_up(torch.float32)
    for n in sel:
        print(f"{n:32s} shared upstream: hip-f64pin {rel_l2(hip[n], u64[n]):.1e}   f32pin-f64pin {rel_l2(u32[n], u64[n]):.1e}")
