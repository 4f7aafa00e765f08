"""Build-container check (no GPU): instruction mix of one kernel of a csrc file -- VALU / SALU / vector-memory counts.
A wave-uniform index computed on the VALU (e.g. from `threadIdx.x >> 6` without readfirstlane) shows up as thousands of
v_ instructions where s_ ones would do.   python tools/isa_mix.py conv_wino.hip wino_in444_kernel"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bodyct-dram-emph-subtype_amd", "csrc")
f, pat = os.path.join(CSRC, sys.argv[1]), sys.argv[2]
out = "/tmp/isa_mix.s"
r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                    '-DDRAM_ABI_HASH="isa"', "--offload-device-only", "-S", f, "-o", out], capture_output=True, text=True)
assert r.returncode == 0, r.stderr[-500:]
txt = open(out).read()
for m in re.finditer(r"\n(_Z\w+):[^\n]*\n(.*?)s_endpgm", txt, re.S):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    if pat not in dem:
        continue
    ins = [l.strip().split()[0] for l in body.split("\n") if l.strip() and l.startswith("\t") and not l.strip().startswith((".", ";"))]
    cnt = lambda p: sum(1 for i in ins if i.startswith(p))
    print(f"{dem[:90]:90s} VALU {cnt('v_'):5d} SALU {cnt('s_'):5d} gload {cnt('global_load'):4d} gstore {cnt('global_store'):4d} "
          f"ds {cnt('ds_'):4d} scratch {cnt('scratch_'):3d} total {len(ins)}")
