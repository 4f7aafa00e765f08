#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 900 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wino or winograd" > $O/t19_k.log 2>&1; echo "kernels rc=$?"; tail -3 $O/t19_k.log
timeout -k 10 900 python -m pytest tests/test_network_gpu.py -x -q -m gpu -k "golden" > $O/t19_n.log 2>&1; echo "golden rc=$?"; tail -3 $O/t19_n.log
for c in "1 f32" "3 f32" "3 bf16" "5 f32"; do set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b19_c$1$2.json 2>$O/b19_c$1$2.err
  python -c "import json;d=json.load(open('$O/b19_c$1$2.json'));print('config $1 $2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager')"
done
python bench.py --config 1 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b19_c1e.json 2>/dev/null
python -c "import json;d=json.load(open('$O/b19_c1e.json'));print('config 1 eager', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
