set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py tests/test_models_gpu.py -x -q -m gpu > gpurun_out/ab/pytest.txt 2>&1 || { tail -40 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
for v in 4 8; do echo "== DRAM_COLREDUCE_VW=$v"; DRAM_COLREDUCE_VW=$v python tools/ew_bench.py 2>&1 | grep "bfloat16.*bn_bwd_reduce"; done
for r in 1 2; do for v in 4 8; do DRAM_COLREDUCE_VW=$v python bench.py --no-cpu-baseline --timeline off --config 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('VW=$v config2', round(d['value'],2), round(d['ms_per_step'],3))"; done; done
