#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for m in "" 1.0; do
  echo "== DRAM_WGRAD_MARGIN=$m"
  for geo in "2 32 64 64 64 64" "2 64 128 128 64 32" "1 32 64 64 64 64"; do
    DRAM_WGRAD_MARGIN=$m python tools/conv_bench.py $geo 3 1 1 wgrad 20 2>&1 | grep -v amdgpu.ids
  done
done
for m in "" 1.0; do
  DRAM_TUNING=1 DRAM_WGRAD_MARGIN=$m python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b21_$m.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b21_$m.json'));print('config 1 margin=$m', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
