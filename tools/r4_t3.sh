#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q -m gpu > $O/t3_kernels.log 2>&1; echo "kernels+models rc=$?"; tail -3 $O/t3_kernels.log
timeout -k 10 600 python -m pytest tests/test_network_gpu.py -x -q -m gpu -k "golden" > $O/t3_net.log 2>&1; echo "golden rc=$?"; tail -3 $O/t3_net.log
for c in "1 f32" "3 f32" "2 bf16" "3 bf16" "0 f32"; do
  set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b3_c$1_$2.json 2>$O/b3_c$1_$2.err
  python -c "import json;d=json.load(open('$O/b3_c$1_$2.json'));print('config $1 $2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager', round(d['peak_hbm_gb'],1),'GB')"
done
python bench.py --config 3 --dtype f32 --graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b3_c3_f32_graph.json 2>/dev/null
python -c "import json;d=json.load(open('$O/b3_c3_f32_graph.json'));print('config 3 f32 graph', round(d['value'],2),'vol/s', round(d['ms_per_step'],3))"
