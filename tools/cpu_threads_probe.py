"""GPU box, CPU only: how many host threads does the fp64 oracle convolution want?  (the box exposes more hardware
threads than the job's CPU share; torch defaults to all of them)"""
import os
import time

import torch
import torch.nn.functional as F

print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
try:
    print("cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip())
except OSError as e:
    print("cpu.max unreadable", e)
x = torch.randn(1, 64, 32, 64, 64, dtype=torch.float64)
w = torch.randn(64, 64, 3, 3, 3, dtype=torch.float64)
for n in (torch.get_num_threads(), 64, 32, 16, 8):
    torch.set_num_threads(n)
    F.conv3d(x, w, None, 1, 1)
    t = time.perf_counter()
    for _ in range(3):
        F.conv3d(x, w, None, 1, 1)
    print(f"threads {n}: {(time.perf_counter() - t) / 3:.3f} s per fp64 64->64 conv at 32x64x64", flush=True)
