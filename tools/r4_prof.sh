#!/bin/bash
# rocprofv3 kernel stats: config 1 (fp32) and config 3 (bf16), single-stream eager steps
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export DRAM_TUNING=1 DRAM_WGRAD_STREAM=0
rm -rf $O/trace_c1 $O/trace_c3bf
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c1 -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --timeline off --no-graph > $O/prof_c1.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3bf -- python3 $R/bench.py --config 3 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --timeline off --no-graph > $O/prof_c3bf.json 2>/dev/null
find $O/trace_c1 $O/trace_c3bf -name "*agent_info.csv" -delete
for t in trace_c1 trace_c3bf; do f=$(find $O/$t -name '*kernel_stats.csv' | head -1); cp $f $O/$t.kernel_stats.csv; done
find $O/trace_c1 $O/trace_c3bf -name "*kernel_trace.csv" -size +20M -delete
ls -la $O/*.kernel_stats.csv
