#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for s in 1 2; do
  DRAM_TUNING=1 DRAM_GRAPH_STREAMS=$s python bench.py --config 1 --graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b23_g$s.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b23_g$s.json'));print('config 1 graph streams=$s', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b23_e.json 2>/dev/null
python -c "import json;d=json.load(open('$O/b23_e.json'));print('config 1 eager', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', d.get('roofline',{}).get('traffic'))"
for s in 1 2; do
  DRAM_TUNING=1 DRAM_GRAPH_STREAMS=$s python bench.py --config 3 --dtype f32 --graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b23_c3g$s.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b23_c3g$s.json'));print('config 3 f32 graph streams=$s', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
