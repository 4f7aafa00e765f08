#!/bin/bash
# rocprofv3 kernel stats: config 2 (bf16), single-stream eager steps
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DRAM_TUNING=1 DRAM_WGRAD_STREAM=0
rm -rf $O/trace_c2bf
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2bf -- python3 $R/bench.py --config 2 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --timeline off --no-graph > $O/prof_c2bf.json 2>/dev/null
find $O/trace_c2bf -name "*agent_info.csv" -delete
f=$(find $O/trace_c2bf -name '*kernel_stats.csv' | head -1); cp $f $O/trace_c2bf.kernel_stats.csv
find $O/trace_c2bf -name "*kernel_trace.csv" -size +20M -delete
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/trace_c2bf.kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print("total ms/step", tot/7e6)
for r in rows[:40]:
    print(f"{float(r['TotalDurationNs'])/7e6:8.3f} {int(r['Calls'])/7:6.1f} {float(r['AverageNs'])/1e3:8.1f}  {r['Name'][:90]}")
PY
