#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "winograd or linearity" > $O/t11_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $O/t11_kernels.log
export DRAM_TUNING=1
for tn in 0 1; do
  echo "== 64->64 @2x64x128x128 pipeline, DRAM_TN64=$tn"
  DRAM_TN64=$tn DRAM_CONV_ALGO=2 python tools/conv_bench.py 2 64 128 128 64 64 3 1 1 wgrad 10 2>&1 | grep -E "^wgrad" | cut -c1-8,90-130
done
unset DRAM_TUNING
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b11_c1.json 2>/dev/null
python -c "import json;d=json.load(open('$O/b11_c1.json'));print('config 1', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
