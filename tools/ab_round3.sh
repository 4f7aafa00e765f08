set -e
mkdir -p gpurun_out/ab
timeout -k 10 400 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu > gpurun_out/ab/pytest.txt 2>&1 || { tail -40 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
timeout -k 10 200 python tools/conv_bf16_bench.py 10 2>&1 | grep -v amdgpu | sed 's/fwd.*wgrad/wgrad/'
for c in 2; do python bench.py --no-cpu-baseline --timeline off --config $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config$c', d['value'], d['ms_per_step'])"; done
python bench.py --no-cpu-baseline --timeline off --config 3 --dtype bf16 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config3 bf16', d['value'], d['ms_per_step'])"
