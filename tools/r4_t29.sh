#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_models_gpu.py tests/test_distributed_gpu.py tests/test_bf16_gpu.py -x -q -m gpu 2>&1 | tail -2
python bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b29_e.json 2>/dev/null
python bench.py --config 3 --dtype bf16 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b29_e3.json 2>/dev/null
python bench.py --config 3 --dtype bf16 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b29_fd3.json 2>/dev/null
python bench.py --config 3 --dtype f32 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b29_e3f.json 2>/dev/null
python bench.py --config 0 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b29_e0.json 2>/dev/null
python - <<PY
import json
for n in ("e","e3","fd3","e3f","e0"):
    d=json.load(open("$O/b29_%s.json"%n))
    print(n, round(d['value'],2), round(d['ms_per_step'],3), 'host', round(d['host_issue_ms_per_step'],2), d.get('exposed_collective_ms'))
PY
