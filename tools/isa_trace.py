"""Order of LDS-DMA issues, waits, LDS reads, barriers, MFMAs, global loads / stores in the gfx950 ISA of a kernel.
   python tools/isa_trace.py bodyct-dram-emph-subtype_amd/csrc/conv_wino2d.hip [kernel-name-substring]

One line per kernel, run-length compressed:  D = global_load_lds, L / S = global load / store, R = ds_read,
M = v_mfma, B = s_barrier, Wn = s_waitcnt vmcnt(n), X = scratch access, | = basic-block boundary.
What to look for (both cost this round's kernels 5-15 % before they were found):
  * "D W0 R" -- a wait for ALL vector memory between the issue of the next stage's DMA and the current stage's
    reads: the double buffering is waited away.  The wait-count pass only lets LDS reads pass an outstanding
    LDS-DMA when alias analysis proves them disjoint: use one __shared__ array per buffer, not one carved array.
  * "| L W0 | L W0 |" -- one guarded load per basic block, waited for one at a time: wave-uniform
    "ok ? p[i] : 0" ternaries become branches; load from a clamped address and select afterwards.
"""
import os
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
with tempfile.TemporaryDirectory() as td:
    out = os.path.join(td, "k.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-I" + os.path.join(root, "include"),
                    "-I" + os.path.dirname(os.path.abspath(src)), "-S", "--cuda-device-only", src, "-o", out],
                   check=True, stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")

cur, seqs = None, {}
for line in lines:
    m = re.match(r"^(_Z\w+):", line)
    if m:
        cur = m.group(1)
        seqs[cur] = []
        continue
    if cur is None:
        continue
    t = line.strip()
    s = seqs[cur]
    if t.startswith("global_load_lds") or (t.startswith("buffer_load") and " lds" in t):
        s.append("D")
    elif t.startswith("global_load") or t.startswith("buffer_load"):
        s.append("L")
    elif t.startswith("global_store") or t.startswith("buffer_store"):
        s.append("S")
    elif t.startswith("scratch_"):
        s.append("X")
    elif t.startswith("s_waitcnt") and "vmcnt" in t:
        s.append("W" + re.search(r"vmcnt\((\d+)\)", t).group(1))
    elif t.startswith("ds_read"):
        s.append("R")
    elif t.startswith("v_mfma"):
        s.append("M")
    elif t.startswith("s_barrier"):
        s.append("B")
    elif re.match(r"^\.LBB", t):
        s.append("|")
    elif t.startswith("s_endpgm"):
        cur = None

for name, toks in seqs.items():
    if want not in name or not toks:
        continue
    res, i = [], 0
    while i < len(toks):
        j = i
        while j < len(toks) and toks[j] == toks[i]:
            j += 1
        res.append(toks[i] + (str(j - i) if j - i > 1 else ""))
        i = j
    flat = re.sub(r"\|\d*", "", " ".join(res))
    bad = len(re.findall(r"D\d* W0 R", flat))
    print(f"{re.sub(r'_ZN12_GLOBAL__N_1[0-9]+', '', name)[:70]}\n   DMA->wait(0)->read: {bad}\n   {' '.join(res)[:1500]}\n")
