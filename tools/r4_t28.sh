#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_models_gpu.py tests/test_distributed_gpu.py -x -q -m gpu 2>&1 | tail -2
python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b28_e.json 2>/dev/null
python bench.py --config 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b28_fd.json 2>/dev/null
python bench.py --config 2 --dtype bf16 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b28_e2.json 2>/dev/null
python bench.py --config 2 --dtype bf16 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b28_fd2.json 2>/dev/null
python bench.py --config 3 --dtype bf16 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b28_e3.json 2>/dev/null
python bench.py --config 3 --dtype bf16 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b28_fd3.json 2>/dev/null
python bench.py --config 0 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b28_e0.json 2>/dev/null
python - <<PY
import json
for n in ("e","fd","e2","fd2","e3","fd3","e0"):
    d=json.load(open("$O/b28_%s.json"%n))
    print(n, round(d['value'],2), round(d['ms_per_step'],3), 'host', round(d['host_issue_ms_per_step'],2), d.get('exposed_collective_ms'))
PY
python tools/host_profile.py 3 bf16 5 > $O/hostprof_c3bf.txt 2>&1
