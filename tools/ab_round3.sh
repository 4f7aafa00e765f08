set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "conv or network" > gpurun_out/ab/pytest.txt 2>&1 || { tail -40 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
timeout -k 10 200 python tools/conv_bf16_bench.py 10 2>&1 | grep -v amdgpu | sed 's/fwd.*wgrad/wgrad/' | grep -E "us3|us2.1|per step"
python bench.py --no-cpu-baseline --timeline off --config 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config2', round(d['value'],2), round(d['ms_per_step'],3))"
