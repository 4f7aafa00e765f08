"""Single-layer micro-benchmark of the conv kernels (HIP events), for A/B tuning.
   python tools/conv_bench.py B D H W Cin Cout k stride dil [fwd|dgrad|wgrad] [iters]"""
import os, sys
os.environ.setdefault("DRAM_TUNING", "1")   # tuning tool: the A/B switches below count
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyct_dram_emph_subtype_amd import ops
import bodyct_dram_emph_subtype_amd as dram

dram.load_library()
B, D, H, W, Cin, Cout, k, stride, dil = [int(v) for v in sys.argv[1:10]]
modes = sys.argv[10].split(",") if len(sys.argv) > 10 else ["fwd", "dgrad", "wgrad"]
iters = int(sys.argv[11]) if len(sys.argv) > 11 else 10
pad = dil * (k - 1) // 2
g = ops.ConvGeom(B, D, H, W, Cin, Cout, k, stride, pad, dil)
dev = "cuda:0"
x = torch.randn(g.in_shape, device=dev)
if os.environ.get("BENCH_ZERO"):      # power / clock probe: all-zero activations toggle far fewer bits
    x.zero_()
w = torch.randn(Cout, Cin, k, k, k, device=dev) * 0.05
dy = torch.randn(g.out_shape, device=dev)
if os.environ.get("BENCH_ZERO"):
    dy.zero_()
wf, wb = ops.pack_conv_weight(w, True, True, g)
print("plan: algo", ops.conv_algo(g))


V = ops.conv3d_fwd_keep(x, wf, None, g, True, True)[2]      # Winograd pipeline: the forward's transformed input
if V is not None:
    print("weight gradient timed WITH the forward's cached Winograd-domain input (as in the train step)")


def run(mode):
    if mode == "fwd":
        return ops.conv3d_fwd(x, wf, None, g, True)
    if mode == "dgrad":
        return ops.conv3d_bwd_data(dy, wb, g)
    return ops.conv3d_bwd_weight(x, dy, g, v_cache=V)


for mode in modes:
    for _ in range(2):
        run(mode)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        run(mode)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / iters
    print(f"{mode:6s} {g}  {ms:8.3f} ms  {g.flops / ms / 1e9:7.1f} TFLOP/s", flush=True)
