set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py -x -q -m gpu > gpurun_out/ab/pytest.txt 2>&1 || { tail -40 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
for w in 512 256 128; do echo "== ZWALK_WGS=$w"; DRAM_BF16_ZWALK_WGS=$w timeout -k 10 200 python tools/conv_bf16_bench.py 10 2>&1 | grep -v amdgpu | sed 's/fwd.*wgrad/wgrad/' | grep -E "layer1|us1.1|us2|us3|per step"; done
timeout -k 10 200 python tools/conv_bf16_bench.py 10 2>&1 | grep -v amdgpu | sed 's/fwd.*wgrad/wgrad/'
for c in 2 1; do python bench.py --no-cpu-baseline --timeline off --config $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config$c', round(d['value'],2), round(d['ms_per_step'],3))"; done
