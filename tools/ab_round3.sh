set -e
mkdir -p gpurun_out/ab
python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py -x -q -m gpu > gpurun_out/ab/pytest.txt 2>&1 || { tail -40 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
python tools/ew_bench.py 2>&1 | grep -v amdgpu.ids > gpurun_out/ab/ew_new.txt
DRAM_POOL_VW=4 python tools/ew_bench.py 2>&1 | grep bfloat16 | grep "pool\|upcat" > gpurun_out/ab/ew_vw4.txt
cat gpurun_out/ab/ew_new.txt; echo "--- VW=4"; cat gpurun_out/ab/ew_vw4.txt
for c in 2 1; do python bench.py --no-cpu-baseline --timeline off --config $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config$c', d['value'], d['ms_per_step'])"; done
