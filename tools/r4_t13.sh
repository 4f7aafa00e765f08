#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "prologue or winograd_path" > $O/t13_k.log 2>&1; echo "k rc=$?"; tail -2 $O/t13_k.log
timeout -k 10 900 python -m pytest tests/test_network_gpu.py tests/test_models_gpu.py tests/test_bf16_gpu.py -x -q -m gpu -k "golden or recompute or graphed or inference or train_step_bf16 or mid_size" > $O/t13_n.log 2>&1; echo "n rc=$?"; tail -3 $O/t13_n.log
export DRAM_TUNING=1
for pr in 0 1; do
 for c in "1 f32" "3 f32"; do set -- $c
  DRAM_BN_PROLOGUE=$pr python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b13_c$1_p$pr.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b13_c$1_p$pr.json'));print('config $1 prologue=$pr', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', round(d['peak_hbm_gb'],1),'GB')"
 done
done
