set -e
mkdir -p gpurun_out/ab
timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu -k "conv or wgrad or weight" > gpurun_out/ab/pytest.txt 2>&1 || { tail -40 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
echo "== old tile form"; DRAM_BF16_WGRAD_TILE=old timeout -k 10 200 python tools/conv_bf16_bench.py 10 2>&1 | grep -v amdgpu
echo "== new tile form"; timeout -k 10 200 python tools/conv_bf16_bench.py 10 2>&1 | grep -v amdgpu
python bench.py --no-cpu-baseline --timeline off --config 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config2', d['value'], d['ms_per_step'])"
