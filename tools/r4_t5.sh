#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_models_gpu.py -x -q -m gpu -k "batchnorm or bn or fold or stats or train or step or graph" > $O/t5_kernels.log 2>&1; echo "kernels+models rc=$?"; tail -2 $O/t5_kernels.log
export DRAM_TUNING=1
for m in 1.15 1.0; do
 for c in "1 f32" "3 f32"; do
  set -- $c
  DRAM_W2D_MARGIN=$m python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b5_c$1_m$m.json 2>$O/b5_c$1_m$m.err
  python -c "import json;d=json.load(open('$O/b5_c$1_m$m.json'));print('config $1 $2 margin $m', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', round(d['peak_hbm_gb'],1),'GB')"
 done
done
unset DRAM_TUNING
bash tools/r4_prof.sh > /dev/null 2>&1
