#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O
cd $R
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --timeline off > $O/b11_c4.json 2>/dev/null
python -c "import json;d=json.load(open('$O/b11_c4.json'));print('config 4', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager', round(d['peak_hbm_gb'],1),'GB')"
python bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b11_c2.json 2>/dev/null
python -c "import json;d=json.load(open('$O/b11_c2.json'));print('config 2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager')"
python bench.py --config 2 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b11_c2e.json 2>/dev/null
python -c "import json;d=json.load(open('$O/b11_c2e.json'));print('config 2 eager', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
cd /tmp && export TMPDIR=/tmp
export DRAM_TUNING=1 DRAM_WGRAD_STREAM=0
rm -rf $O/trace_c2
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python3 $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --timeline off --no-graph > $O/prof_c2.json 2>/dev/null
f=$(find $O/trace_c2 -name '*kernel_stats.csv' | head -1); cp $f $O/trace_c2.kernel_stats.csv
find $O/trace_c2 -name "*kernel_trace.csv" -delete; find $O/trace_c2 -name "*agent_info.csv" -delete
