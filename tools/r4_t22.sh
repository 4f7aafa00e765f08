#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for c in "1 f32" "0 f32" "3 f32" "5 f32" "2 f32" "2 bf16" "3 bf16"; do set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b22_c$1$2.json 2>$O/b22_c$1$2.err
  python -c "import json;d=json.load(open('$O/b22_c$1$2.json'));print('config $1 $2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager')"
done
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu --durations=15 > $O/full.log 2>&1; rc=$?
tail -30 $O/full.log
exit $rc
