#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "segloss or tail or fold or reduce or partial or batchnorm or bn" > $O/t18_k.log 2>&1; echo "kernels rc=$?"; tail -3 $O/t18_k.log
timeout -k 10 600 python -m pytest tests/test_models_gpu.py tests/test_network_gpu.py -x -q -m gpu -k "mid_size or models or graph or trainer" > $O/t18_n.log 2>&1; echo "net rc=$?"; tail -3 $O/t18_n.log
for c in "2 bf16" "1 fp32"; do set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b18_c$1.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b18_c$1.json'));print('config $1 $2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager')"
  python bench.py --config $1 --dtype $2 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b18_c$1e.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b18_c$1e.json'));print('config $1 $2 eager', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
