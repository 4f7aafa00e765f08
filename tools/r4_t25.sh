#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b25_e.json 2>/dev/null
python bench.py --config 1 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b25_fd.json 2>/dev/null
python bench.py --config 2 --dtype bf16 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b25_e2.json 2>/dev/null
python bench.py --config 2 --dtype bf16 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b25_fd2.json 2>/dev/null
wc -l $O/b25_*.json
python - <<PY
import json
for n in ("e","fd","e2","fd2"):
    d=json.load(open("$O/b25_%s.json"%n))
    print(n, round(d['value'],2), round(d['ms_per_step'],3), 'host', round(d['host_issue_ms_per_step'],2), d.get('exposed_collective_ms'))
PY
timeout -k 10 300 python -m pytest tests/test_distributed_gpu.py -x -q -m gpu -k "bench_gpus2" 2>&1 | tail -2
