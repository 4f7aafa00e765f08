#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu > $O/t20_b.log 2>&1; echo "bf16 rc=$?"; tail -5 $O/t20_b.log
for c in "2 bf16" "3 bf16"; do set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b20_c$1.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b20_c$1.json'));print('config $1 $2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager')"
  python bench.py --config $1 --dtype $2 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b20_c$1e.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b20_c$1e.json'));print('config $1 $2 eager', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
