"""Build-container check (no GPU): compile every csrc/*.hip to gfx950 assembly and list, per kernel,
  * LDS-DMA instructions, scratch (spill) traffic,
  * `s_waitcnt vmcnt(N)` instructions that are NOT directly in front of an `s_barrier` and sit behind an LDS-DMA issue:
    the wait-count pass put them there because it could not tell the LDS object an operand read touches from the target
    of a pending DMA (a selected buffer pointer, a run-time ring slot), or because an inline-asm wait hid the fact that
    the DMA had already been waited for, or because a spilled register is reloaded (scratch loads count on vmcnt).
    Each one is a point where the kernel waits for the prefetch it has just issued.
   python tools/isa_waits.py [--table] [file.hip ...]        (inline-asm waits are marked `asm`)
--table: VGPRs, LDS bytes and spills of EVERY kernel instead -- look for LDS in kernels that declare none (a private
array indexed by a loop variable is moved to LDS by the compiler: upcat_bwd_src_kernel carried 56 KB per workgroup
that way) and for register counts far above the live values (head_fwd_kernel<9>: 366 VGPRs of hoisted weight reads)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bodyct-dram-emph-subtype_amd", "csrc")
TABLE = "--table" in sys.argv
files = [a for a in sys.argv[1:] if a != "--table"] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def short_name(name):
    dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    mm = re.search(r"(\w+(?:<[^>]*>)?)\((?!anonymous)", dem)
    return (mm.group(1) if mm else dem)[:70]


is_dma = lambda l: "global_load_lds" in l or ("buffer_load" in l and l.rstrip().endswith("lds"))   # noqa: E731
for f in files:
    out = f"/tmp/isa_{os.path.basename(f)}.s"
    r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-I", CSRC,
                        "-DDRAM_ABI_HASH=\"isa\"", "--offload-device-only", "-S", f, "-o", out], capture_output=True, text=True)
    if r.returncode:
        print(f"{f}: compile failed\n{r.stderr[-400:]}")
        continue
    txt = open(out).read()
    if TABLE:
        for m in re.finditer(r"\.group_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n"
                             r"\s+\.vgpr_spill_count:\s+(\d+)", txt):
            lds, name, vg, sp = m.groups()
            print(f"{os.path.basename(f):18s} {short_name(name):70s} vgpr {vg:>4s}  lds {lds:>6s}  spills {sp}")
        continue
    spills = dict((m.group(1), int(m.group(2))) for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", txt))
    for m in re.finditer(r"^(_Z[^\n:]*):.*?s_endpgm", txt, re.S | re.M):
        name, body = m.group(1), m.group(0).split("\n")
        ndma = sum(is_dma(l) for l in body)
        if not ndma and not spills.get(name):
            continue
        inasm, susp = False, []
        for i, l in enumerate(body):
            inasm = True if "#ASMSTART" in l else (False if "#ASMEND" in l else inasm)
            w = re.search(r"s_waitcnt.*vmcnt\((\d+)\)", l)
            if not w:
                continue
            nxt = [b for b in body[i + 1:i + 6] if b.strip() and not b.strip().startswith(";")]
            if any("s_barrier" in x for x in nxt[:3]):
                continue
            if any(is_dma(x) for x in body[max(0, i - 80):i]):
                susp.append(f"{i}:{'asm' if inasm else 'cc'}:vmcnt({w.group(1)})")
        short = short_name(name)
        print(f"{os.path.basename(f):18s} {short:70s} dma {ndma:3d}  spills {spills.get(name, 0):3d}  waits behind a DMA, not at a barrier: {susp[:8]}")
