"""Turns the raw output of tools/profile_round.sh into the committed profiles/rNN_* files.
   python tools/summarize_profile.py gpurun_out/prof profiles r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)

FAMILIES = [("conv_wino2d_kernel", "conv_wino2d"), ("wino_gemm_nn", "wino_gemm_nn"), ("wino_gemm_tn", "wino_gemm_tn"),
            ("wino_in_kernel", "wino_transforms"), ("wino_in444", "wino_transforms"), ("wino_out_kernel", "wino_transforms"),
            ("wino_wgrad_out", "wino_transforms"), ("wino_weight", "weight_pack"), ("wino2d_weight", "weight_pack"),
            ("pack_weight", "weight_pack"), ("conv_igemm", "conv_igemm"), ("conv_wgrad", "conv_wgrad"),
            ("wgrad_reduce", "conv_wgrad"), ("stem_", "stem"), ("bn_", "bn/elementwise"), ("colreduce", "bn/elementwise"),
            ("reduce_partials", "bn/elementwise"), ("upcat", "bn/elementwise"), ("maxpool", "bn/elementwise"),
            ("add_kernel", "bn/elementwise")]


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "other"


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")


for f in glob.glob(os.path.join(src, "bench_config*")):
    shutil.copy(f, os.path.join(dst, f"{tag}_" + os.path.basename(f)))

# ---- kernel trace summary -----------------------------------------------------------------
stats = max(glob.glob(os.path.join(src, "trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)   # newest run
shutil.copy(stats, os.path.join(dst, f"{tag}_rocprofv3_kernel_stats_bench_config1.csv"))
rows = list(csv.DictReader(open(stats)))
steps = 7.0   # 2 warm-up + 5 timed
tot = sum(float(r["TotalDurationNs"]) for r in rows)
fam = collections.defaultdict(lambda: [0.0, 0])
with open(os.path.join(dst, f"{tag}_rocprofv3_summary.txt"), "w") as out:
    out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 "
              "--no-cpu-baseline   (config 1, one MI355X)\n")
    out.write(f"# total kernel time {tot / steps / 1e6:.2f} ms/step over 7 steps (2 warm-up + 5 timed); "
              "columns: ms/step, calls/step, avg us, %, kernel\n")
    for r in rows[:48]:
        out.write(f"{float(r['TotalDurationNs']) / steps / 1e6:9.3f} {float(r['Calls']) / steps:7.1f} "
                  f"{float(r['AverageNs']) / 1e3:10.1f} {float(r['Percentage']):6.2f}  {short(r['Name'])[:140]}\n")
    for r in rows:
        fm = family(r["Name"])
        fam[fm][0] += float(r["TotalDurationNs"])
        fam[fm][1] += int(r["Calls"])
    out.write("\n")
    for k, (ns, calls) in sorted(fam.items(), key=lambda kv: -kv[1][0]):
        out.write(f"# family {k}: {ns / steps / 1e6:.3f} ms/step, {calls / steps:.1f} launches/step, "
                  f"average launch {ns / max(calls, 1) / 1e3:.1f} us\n")
    try:
        line = [l for l in open(os.path.join(src, "bench_config1.json.log")) if l.startswith("{")][-1]
        j = json.loads(line)
        rf = j["roofline"]
        out.write(f"# bench.py (same build, no profiler): {j['value']:.2f} volumes/s, {j['ms_per_step']:.2f} ms/step; "
                  f"roofline kernel {rf['kernel']}: avg launch {rf['avg_launch_ms'] * 1e3:.1f} us, "
                  f"{rf['achieved']:.1f} TFLOP/s algorithmic = {rf['frac']:.3f} of {rf['peak']}\n")
    except Exception as e:  # noqa: BLE001
        out.write(f"# (bench line not parsed: {e})\n")

# ---- PMC passes -----------------------------------------------------------------------------
def pmc(dirname):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(set))
    for f in sorted(glob.glob(os.path.join(src, dirname, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
        for r in csv.DictReader(open(f)):
            fm = family(r["Kernel_Name"])
            acc[fm][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[fm][r["Counter_Name"]].add(r["Dispatch_Id"])
    return acc, launches


fetch, lf = pmc("pmc_fetch")
write, lw = pmc("pmc_write")
traffic = {}
for fm in sorted(set(fetch) | set(write)):
    n = max(len(lf[fm].get("FETCH_SIZE", ())), len(lw[fm].get("WRITE_SIZE", ())), 1)
    # FETCH_SIZE / WRITE_SIZE are reported in KiB; gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2
    rd = fetch[fm].get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 / n
    wr = write[fm].get("WRITE_SIZE", 0.0) * 1024.0 / n
    traffic[fm] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "total": rd + wr}
json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic.json"), "w"), indent=1)

sq, ls = pmc("pmc_sq")
with open(os.path.join(dst, f"{tag}_pmc_sq_counters.txt"), "w") as out:
    out.write("# rocprofv3 --pmc SQ_* (one pass, kernel-trace only) -- python bench.py --steps 3 --warmup 1; sums over all "
              "launches of the family\n")
    for fm, d in sorted(sq.items()):
        out.write(fm + "\n")
        for c, v in sorted(d.items()):
            out.write(f"   {c:28s} {v:.4g}\n")
print("wrote", sorted(os.listdir(dst)))
