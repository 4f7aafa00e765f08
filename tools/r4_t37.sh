#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for i in 1 2; do
for v in 1 0; do
  DRAM_TUNING=1 DRAM_BN_PROLOGUE=$v python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b37_c1_$v.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b37_c1_$v.json'));print('config 1 prologue=$v', round(d['value'],2), round(d['ms_per_step'],3), round(d['peak_hbm_gb'],1))"
done; done
for v in 1 0; do
  DRAM_TUNING=1 DRAM_BN_PROLOGUE=$v python bench.py --config 3 --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b37_c3_$v.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b37_c3_$v.json'));print('config 3 prologue=$v', round(d['value'],2), round(d['ms_per_step'],3), round(d['peak_hbm_gb'],1))"
done
