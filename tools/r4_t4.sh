#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "batchnorm or bn or fold or stats or train or step or graph" > $O/t4_kernels.log 2>&1; echo "kernels+models rc=$?"; tail -3 $O/t4_kernels.log
timeout -k 10 600 python -m pytest tests/test_network_gpu.py -x -q -m gpu -k "golden and (net_0 or net_6 or net_3)" > $O/t4_net.log 2>&1; echo "golden rc=$?"; tail -3 $O/t4_net.log
bash tools/r4_prof.sh > /dev/null 2>&1
cd $R
for c in "1 f32" "3 f32" "2 bf16" "3 bf16"; do
  set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b4_c$1_$2.json 2>$O/b4_c$1_$2.err
  python -c "import json;d=json.load(open('$O/b4_c$1_$2.json'));print('config $1 $2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager')"
done
