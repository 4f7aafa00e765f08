#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the data-parallel step at world size 1 with every collective in place
# (bench.py --force-dist) for configs 1 (fp32), 2 (bf16) and 3 in bf16, next to the plain step of the same process
# (plain_ms_per_step in each line).  Output: gpurun_out/dp/; stderr of every command kept next to its output.
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/dp; mkdir -p $O
cd $R
for spec in "1 f32" "2 bf16" "3 bf16"; do
  set -- $spec
  python bench.py --config $1 --dtype $2 --force-dist --steps 20 --warmup 5 --no-cpu-baseline --timeline off \
      > $O/fd_c$1_$2.json 2> $O/fd_c$1_$2.err || echo "config $1 $2: rc=$?" >> $O/failures.txt
done
if [ "$1" != "" ] && [ "${DP_AB:-0}" = "1" ]; then
  # A/B: the same collectives through torch.distributed's ProcessGroupNCCL (round 4's transport)
  export DRAM_TUNING=1 DRAM_DIST_TRANSPORT=torch
  for spec in "1 f32" "2 bf16" "3 bf16"; do
    set -- $spec
    python bench.py --config $1 --dtype $2 --force-dist --no-graph --steps 20 --warmup 5 --no-cpu-baseline --timeline off \
        > $O/fd_torch_c$1_$2.json 2> $O/fd_torch_c$1_$2.err || echo "torch transport config $1 $2: rc=$?" >> $O/failures.txt
  done
fi
python - <<PY
import json, glob
for f in sorted(glob.glob("$O/fd_*.json")):
    try:
        r = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable:", e); continue
    print(f.split("/")[-1], "ms/step", round(r["ms_per_step"], 2), "plain", round(r.get("plain_ms_per_step", 0), 2),
          "exposed", round(r.get("exposed_collective_ms", 0), 2), "host_issue", round(r["host_issue_ms_per_step"], 2),
          "graph", r["config"]["hip_graph"], r.get("collective_transport"))
PY
