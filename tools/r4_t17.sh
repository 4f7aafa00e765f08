#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests/test_network_gpu.py -x -q -m gpu -s -k "full_size or inference or decisions" > $O/t17_net.log 2>&1; echo "net rc=$?"; grep -E "^\[config|passed|failed|max-rel|AssertionError" $O/t17_net.log | cut -c1-220
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py tests/test_models_gpu.py -x -q -m gpu > $O/t17_b.log 2>&1; echo "bf16+models rc=$?"; tail -2 $O/t17_b.log
for c in "2 bf16" "3 bf16"; do set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b17_c$1.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b17_c$1.json'));print('config $1 $2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager')"
  python bench.py --config $1 --dtype $2 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b17_c$1e.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b17_c$1e.json'));print('config $1 $2 eager', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
