#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "conv1x1_bf16 or train_step_bf16" > $O/t16_k.log 2>&1; echo "k rc=$?"; tail -3 $O/t16_k.log
export DRAM_TUNING=1
for w in small big; do
  DRAM_BF16_WGRAD1=$w python bench.py --config 3 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b16_$w.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b16_$w.json'));print('config 3 bf16 wgrad1=$w', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
unset DRAM_TUNING
bash tools/r4_prof.sh > /dev/null 2>&1
