"""Does the 256 MiB Infinity Cache absorb a producer -> consumer round trip?  Streams y = 2x then z = y + 1 over
buffers of growing size and prints the byte rate of each size (HIP events, torch element-wise kernels)."""
import torch
dev = "cuda:0"
for mb in (8, 16, 32, 48, 64, 96, 128, 192, 256, 512, 1024, 2048):
    n = mb * (1 << 20) // 4
    x = torch.randn(n, device=dev)
    y = torch.empty_like(x)
    z = torch.empty_like(x)
    for _ in range(3):
        torch.mul(x, 2.0, out=y); torch.add(y, 1.0, out=z)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    it = max(10, 4096 // mb)
    a.record()
    for _ in range(it):
        torch.mul(x, 2.0, out=y); torch.add(y, 1.0, out=z)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / it
    print(f"{mb:5d} MiB per buffer (3 buffers): {ms*1e3:8.1f} us per pair, {4 * n * 4 / ms / 1e6:7.0f} GB/s read+write", flush=True)
