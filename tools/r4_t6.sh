#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "streaming or winograd_path or linearity" > $O/t6_kernels.log 2>&1; echo "kernels rc=$?"; tail -2 $O/t6_kernels.log
export DRAM_TUNING=1
for spec in "2 64 128 128 64 64 3 1 1" "2 64 128 128 128 64 3 1 1"; do
  for st in 0 1; do
    echo "== $spec pipeline, DRAM_NN_STREAM=$st"
    DRAM_NN_STREAM=$st DRAM_CONV_ALGO=2 python tools/conv_bench.py $spec fwd,dgrad 10 2>&1 | grep -E "^fwd|^dgrad" | cut -c1-8,90-130
  done
done
unset DRAM_TUNING
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b6_c1.json 2>/dev/null
python -c "import json;d=json.load(open('$O/b6_c1.json'));print('config 1', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
