#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DRAM_TUNING=1 DRAM_WGRAD_STREAM=0
rm -rf $O/trace_c3
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 $R/bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --timeline off --no-graph > $O/prof_c3.json 2>/dev/null
f=$(find $O/trace_c3 -name '*kernel_stats.csv' | head -1); cp $f $O/trace_c3.kernel_stats.csv
find $O/trace_c3 -name "*kernel_trace.csv" -delete; find $O/trace_c3 -name "*agent_info.csv" -delete
