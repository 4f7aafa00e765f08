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

# kernel name fragment -> family, the library's own taxonomy (include/dram_hip.h DRAM_FAM_*, the names in
# bench.py's roofline.families); first match wins
FAMILIES = [("conv3_bf16", "conv_bf16"), ("gemm1_bf16", "conv_bf16"), ("wgrad3", "wgrad_bf16"), ("wgrad1_bf16", "wgrad_bf16"),
            ("wgrad1b_bf16", "wgrad_bf16"), ("s2d_bf16", "pool_up"), ("d2s_bf16", "pool_up"), ("s2_embed", "weight_pack"),
            ("s2_extract", "weight_pack"), ("regloss", "head_loss"),
            ("cast_", "bn_elementwise"), ("conv_wino2d", "conv_wino2d"), ("wino_in", "wino_in"), ("wino_out_kernel", "wino_out"),
            ("wino_gemm_nn", "wino_gemm_nn"), ("wino_gemm_tn", "wino_gemm_tn"), ("wino_wgrad_out", "wino_wgrad_out"),
            ("slab_sum", "wino_wgrad_out"), ("wino_weight", "weight_pack"), ("wino2d_weight", "weight_pack"),
            ("pack_weight", "weight_pack"), ("conv_wgrad_w2d", "conv_wgrad_w2d"), ("wgrad_w2d_reduce", "conv_wgrad_w2d"),
            ("conv_igemm", "conv_igemm"), ("conv_wgrad", "conv_wgrad"), ("wgrad_reduce", "conv_wgrad"),
            ("stem_", "stem"), ("bn_", "bn_elementwise"), ("colreduce", "bn_elementwise"),
            ("reduce_partials", "bn_elementwise"), ("fold_partials", "bn_elementwise"), ("add_kernel", "bn_elementwise"),
            ("upmix_axis", "pool_up"), ("upmix_split", "weight_pack"), ("upmix_merge", "weight_pack"), ("maxpool", "pool_up"),
            ("upcat", "pool_up"), ("upproject", "pool_up"), ("head_", "head_loss"), ("segloss", "head_loss"),
            ("adam_multi", "optim"), ("sgd_multi", "optim"), ("window_stats", "prep"), ("prep_", "prep")]


def family(name):
    for key, fam in FAMILIES:
        if key in name:
            return fam
    return "torch_glue"


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")


for f in glob.glob(os.path.join(src, "bench_config*")):
    shutil.copy(f, os.path.join(dst, f"{tag}_" + os.path.basename(f)))

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bodyct_dram_emph_subtype_amd import _build  # noqa: E402


def one_config(cfg, dsfx, fsfx):
    """cfg: bench config id; dsfx: suffix of the raw trace / pmc directories; fsfx: suffix of the committed files"""
    if not glob.glob(os.path.join(src, "trace" + dsfx, "*", "*kernel_stats.csv")):
        return
    # ---- kernel trace summary -----------------------------------------------------------------
    stats = max(glob.glob(os.path.join(src, "trace" + dsfx, "*", "*kernel_stats.csv")), key=os.path.getmtime)   # newest run
    shutil.copy(stats, os.path.join(dst, f"{tag}_rocprofv3_kernel_stats_bench_config{cfg}{'_bf16' if 'bf16' in fsfx else ''}.csv"))
    rows = list(csv.DictReader(open(stats)))
    steps = 7.0   # 2 warm-up + 5 timed
    PMC_STEPS = 4.0   # the PMC passes run --steps 3 --warmup 1
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    fam = collections.defaultdict(lambda: [0.0, 0])
    with open(os.path.join(dst, f"{tag}_rocprofv3_summary{fsfx}.txt"), "w") as out:
        out.write("# rocprofv3 --kernel-trace --stats --output-format csv -- python bench.py --steps 5 --warmup 2 --timeline off "
                  f"--no-cpu-baseline --no-graph --config {cfg}{' --dtype bf16' if 'bf16' in fsfx else ''}   "
                  "(one MI355X; DRAM_TUNING=1 DRAM_WGRAD_STREAM=0: single-stream pass)\n")
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
            line = [l for l in open(os.path.join(src, f"bench_config{cfg}{'_bf16' if 'bf16' in fsfx else ''}.json.log")) if l.startswith("{")][-1]
            j = json.loads(line)
            rf = j["roofline"]
            out.write(f"# bench.py (same build, no rocprofv3): {j['value']:.2f} volumes/s, {j['ms_per_step']:.2f} ms/step; "
                      f"roofline family {rf['family']}: avg launch {rf['avg_launch_ms'] * 1e3:.1f} us, "
                      f"{rf['achieved']:.1f} {rf['unit']} = {rf['frac']:.3f} of {rf['peak']}\n")
            out.write("# bench.py kernel timeline (hipEvent pairs), per family: ms/step, launches/step, avg us, bound, frac\n")
            for k, r in sorted(rf["families"].items(), key=lambda kv: -kv[1]["ms_per_step"]):
                out.write(f"#   {k:16s} {r['ms_per_step']:8.3f} {r['launches_per_step']:7.1f} {r['avg_launch_ms'] * 1e3:9.1f}  "
                          f"{r['bound']:4s} {r['frac']:.3f}\n")
        except Exception as e:  # noqa: BLE001
            out.write(f"# (bench line not parsed: {e})\n")

    # ---- PMC passes -----------------------------------------------------------------------------
    def pmc(dirname):
        acc = collections.defaultdict(lambda: collections.defaultdict(float))
        launches = collections.defaultdict(lambda: collections.defaultdict(set))
        for f in sorted(glob.glob(os.path.join(src, dirname + dsfx, "*", "*counter_collection.csv")), key=os.path.getmtime)[-1:]:
            for r in csv.DictReader(open(f)):
                fm = family(r["Kernel_Name"])
                acc[fm][r["Counter_Name"]] += float(r["Counter_Value"])
                launches[fm][r["Counter_Name"]].add(r["Dispatch_Id"])
        return acc, launches


    fetch, lf = pmc("pmc_fetch")
    write, lw = pmc("pmc_write")
    if not fetch and not write:          # trace-only configuration (config 3): no counter passes were run
        return
    traffic = {}
    for fm in sorted(set(fetch) | set(write)):
        n = max(len(lf[fm].get("FETCH_SIZE", ())), len(lw[fm].get("WRITE_SIZE", ())), 1)
        # FETCH_SIZE / WRITE_SIZE are reported in KiB; gfx950: FETCH_SIZE counts 128-B requests as 64 B -> x2
        rd = fetch[fm].get("FETCH_SIZE", 0.0) * 1024.0 * 2.0 / n
        wr = write[fm].get("WRITE_SIZE", 0.0) * 1024.0 / n
        traffic[fm] = {"launches": n, "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "total": rd + wr}
    traffic["_step_total_bytes"] = sum(v["launches"] * v["total"] for v in traffic.values()) / PMC_STEPS
    traffic["source_hash"] = _build.source_hash()     # bench.py ignores this file when the kernel sources differ
    traffic["_note"] = ("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 3 --warmup 1, config " + str(cfg) + "; "
                        "FETCH_SIZE x 2 on gfx950 (128-B requests counted as 64 B); per-launch averages per family")
    json.dump(traffic, open(os.path.join(dst, f"{tag}_pmc_hbm_traffic{fsfx}.json"), "w"), indent=1)

    sq, ls = pmc("pmc_sq")
    with open(os.path.join(dst, f"{tag}_pmc_sq_counters{fsfx}.txt"), "w") as out:
        out.write("# rocprofv3 --pmc SQ_* (one pass, kernel-trace only) -- python bench.py --steps 3 --warmup 1; sums over all "
                  "launches of the family\n")
        for fm, d in sorted(sq.items()):
            out.write(fm + "\n")
            for c, v in sorted(d.items()):
                out.write(f"   {c:28s} {v:.4g}\n")


one_config(1, "", "")
one_config(2, "_c2", "_config2")
one_config(3, "_c3", "_config3")              # ResNet-50 (the reference's default model): kernel traces only
one_config(3, "_c3bf", "_config3_bf16")
print("wrote", sorted(os.listdir(dst)))
