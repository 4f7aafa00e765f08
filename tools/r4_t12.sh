#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "maxpool or one_pass" > $O/t12_k.log 2>&1; echo "k rc=$?"; tail -2 $O/t12_k.log
timeout -k 10 600 python -m pytest tests/test_network_gpu.py tests/test_bf16_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "golden and (net_0 or net_6) or train_step_bf16 or graphed or recompute" > $O/t12_n.log 2>&1; echo "n rc=$?"; tail -2 $O/t12_n.log
for c in "1 f32" "2 bf16"; do set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b12_c$1.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b12_c$1.json'));print('config $1', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
