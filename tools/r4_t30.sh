#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for i in 1 2; do
python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b30_e$i.json 2>/dev/null
DRAM_TUNING=1 python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b30_t$i.json 2>/dev/null
done
python bench.py --config 2 --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b30_c2.json 2>/dev/null
python - <<PY
import json
for n in ("e1","t1","e2","t2","c2"):
    d=json.load(open("$O/b30_%s.json"%n))
    print(n, round(d['value'],2), round(d['ms_per_step'],3), 'host', round(d['host_issue_ms_per_step'],2))
PY
python tools/host_profile.py 1 f32 5 > $O/hostprof_c1b.txt 2>&1
head -30 $O/hostprof_c1b.txt | cut -c1-140
