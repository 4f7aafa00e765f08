#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for i in 1 2; do
python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b31_e$i.json 2>/dev/null
DRAM_TUNING=1 DRAM_INFLIGHT=0 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b31_t$i.json 2>/dev/null
DRAM_TUNING=1 DRAM_INFLIGHT=2 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b31_u$i.json 2>/dev/null
done
python bench.py --config 2 --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b31_c2.json 2>/dev/null
python bench.py --config 3 --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b31_c3.json 2>/dev/null
python bench.py --config 3 --dtype bf16 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b31_c3b.json 2>/dev/null
python bench.py --config 2 --dtype bf16 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b31_c2b.json 2>/dev/null
python bench.py --config 0 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b31_c0.json 2>/dev/null
python - <<PY
import json
for n in ("e1","t1","u1","e2","t2","u2","c2","c3","c3b","c2b","c0"):
    d=json.load(open("$O/b31_%s.json"%n))
    print(n, round(d['value'],2), round(d['ms_per_step'],3), 'host', round(d['host_issue_ms_per_step'],2), 'peak', round(d['peak_hbm_gb'],1))
PY
