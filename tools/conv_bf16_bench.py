"""GPU box: per-layer rate of the bf16-storage convolution kernels (HIP events) on the 3x3x3 stride-1 layers of
BASELINE configs[2] (ResNet-18 + dRAM head, batch 2, 1x128x256x256).   python tools/conv_bf16_bench.py [iters]"""
import os
os.environ.setdefault("DRAM_TUNING", "1")   # tuning tool: the A/B switches below count
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyct_dram_emph_subtype_amd import ops  # noqa: E402
import bodyct_dram_emph_subtype_amd as dram  # noqa: E402

dram.load_library()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
LAYERS = [  # B, D, H, W, Cin, Cout, dil, count
    (2, 32, 64, 64, 64, 64, 1, 4, "layer1"),
    (2, 16, 32, 32, 128, 128, 1, 3, "layer2"),
    (2, 16, 32, 32, 128, 256, 2, 1, "layer3.0.conv1"),
    (2, 16, 32, 32, 256, 256, 2, 3, "layer3"),
    (2, 16, 32, 32, 256, 512, 4, 1, "layer4.0.conv1"),
    (2, 16, 32, 32, 512, 512, 4, 3, "layer4"),
    (2, 32, 64, 64, 576, 64, 1, 1, "us1.0"),
    (2, 32, 64, 64, 64, 64, 1, 1, "us1.1"),
    (2, 64, 128, 128, 128, 64, 1, 1, "us2.0"),
    (2, 64, 128, 128, 64, 64, 1, 1, "us2.1"),
    (2, 64, 128, 128, 64, 32, 1, 1, "us3"),
]
dev = "cuda:0"
tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
with ops.launch_scope(dev):
    for B, D, H, W, Cin, Cout, dil, cnt, name in LAYERS:
        g = ops.ConvGeom(B, D, H, W, Cin, Cout, 3, 1, dil, dil)
        x = torch.randn(g.in_shape, device=dev).bfloat16()
        dy = torch.randn(g.out_shape, device=dev).bfloat16()
        w = torch.randn(Cout, Cin, 3, 3, 3, device=dev) * 0.05
        if os.environ.get("BENCH_ZERO") == "act":      # power / clock probe: all-zero operands toggle far fewer bits
            x.zero_(); dy.zero_()
        elif os.environ.get("BENCH_ZERO") == "w":
            w.zero_()
        wf, wb = ops.pack_conv_weight(w, True, True, g, torch.bfloat16)
        fns = {"fwd": lambda: ops.conv3d_fwd_keep(x, wf, None, g, True, False),
               "dgrad": lambda: ops.conv3d_bwd_data(dy, wb, g),
               "wgrad": lambda: ops.conv3d_bwd_weight(x, dy, g)}
        line = f"{name:16s} {Cin:4d}->{Cout:4d} d{dil} @{D}x{H}x{W}:"
        for mode, fn in fns.items():
            fn(); fn()
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(iters):
                fn()
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b) / iters
            tot[mode] += ms * cnt
            line += f"  {mode} {ms:6.3f} ms {g.flops / ms / 1e9:6.0f} TF"
        print(line, flush=True)
        del x, dy
print("per step (x count):", {k: round(v, 2) for k, v in tot.items()})
