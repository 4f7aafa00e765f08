#!/bin/bash
# round-4 A/B: low-resolution mixing on / off, configs 1, 2, 3 (fp32 + bf16)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
export DRAM_TUNING=1
for c in "1 f32" "3 f32" "3 bf16" "2 bf16"; do
  set -- $c
  for u in 0 1; do
    DRAM_UPMIX=$u python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b1_c$1_$2_u$u.json 2>$O/b1_c$1_$2_u$u.err
    python -c "import json;d=json.load(open('$O/b1_c$1_$2_u$u.json'));print('config $1 $2 upmix=$u', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager', round(d['peak_hbm_gb'],1),'GB')"
  done
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --detail $O/b1_c1_layers.txt > $O/b1_c1_detail.json 2>/dev/null
python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --detail $O/b1_c3_layers.txt > $O/b1_c3_detail.json 2>/dev/null
python bench.py --config 3 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --detail $O/b1_c3bf_layers.txt > $O/b1_c3bf_detail.json 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/b1_c1_timeline.json 2>/dev/null
python -c "import json;d=json.load(open('$O/b1_c1_timeline.json'));w=d['roofline']['whole_step'];print('timeline: step', d['ms_per_step'], 'with timeline', w['ms_per_step_with_timeline'], 'kernel sum', w['kernel_ms_per_step'])"
