"""Rounding error of the Winograd pipeline per math mode (DRAM_MATH) and tiling against an fp64 convolution.
   python tools/math_check.py  [B D H W Cin Cout dil]"""
import os, sys
os.environ.setdefault("DRAM_TUNING", "1")   # tuning tool: the A/B switches below count
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyct_dram_emph_subtype_amd import ops
import bodyct_dram_emph_subtype_amd as dram

dram.load_library()
args = [int(v) for v in sys.argv[1:8]] if len(sys.argv) >= 8 else [1, 8, 16, 16, 256, 256, 1]
B, D, H, W, Cin, Cout, dil = args
dev = "cuda:0"
gen = torch.Generator().manual_seed(0)
x = torch.randn(B, Cin, D, H, W, generator=gen)
w = torch.randn(Cout, Cin, 3, 3, 3, generator=gen) * 0.05
gy = torch.randn(B, Cout, D, H, W, generator=gen)
xr = x.double().to(dev).requires_grad_(True)
wr = w.double().to(dev).requires_grad_(True)
yr = F.conv3d(xr, wr, None, 1, dil, dil)
gxr, gwr = torch.autograd.grad(yr, [xr, wr], gy.double().to(dev))
nd = lambda t: t.permute(0, 2, 3, 4, 1).contiguous().to(dev)
nc = lambda t: t.permute(0, 4, 1, 2, 3)
rel = lambda a, b: float((a.double() - b).norm() / b.norm())
g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, dil, dil)
os.environ["DRAM_CONV_ALGO"] = "2"
for tiling in ["2,2,2", "4,2,2", "4,4,2", "4,4,4"]:
    os.environ["DRAM_WINO_TILING"] = tiling
    for math in ["f32", "bf16x3", "bf16"]:
        os.environ["DRAM_MATH"] = math
        wf, wb = ops.pack_conv_weight(w.to(dev), True, True, g)
        y, _, v = ops.conv3d_fwd_keep(nd(x), wf, None, g, False, True)
        dx = ops.conv3d_bwd_data(nd(gy), wb, g)
        dw = ops.conv3d_bwd_weight(nd(x), nd(gy), g)
        dwc = ops.conv3d_bwd_weight(nd(x), nd(gy), g, v_cache=v)
        print(f"F{tiling.replace(',', '')} {math:7s} fwd {rel(nc(y), yr.detach()):.2e}  dgrad {rel(nc(dx), gxr):.2e}  "
              f"wgrad {rel(dw, gwr):.2e}  wgrad(cached V) {rel(dwc, gwr):.2e}", flush=True)
