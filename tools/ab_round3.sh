set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py tests/test_models_gpu.py -x -q -m gpu > gpurun_out/ab/pytest.txt 2>&1 || { tail -40 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
for v in gather sep; do echo "== DRAM_UPCAT_BWD=$v"; DRAM_UPCAT_BWD=$v python tools/ew_bench.py 2>&1 | grep upcat_bwd; done
for r in 1 2; do for v in gather sep; do for c in 1 2; do DRAM_UPCAT_BWD=$v python bench.py --no-cpu-baseline --timeline off --config $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v config$c', round(d['value'],2), round(d['ms_per_step'],3))"; done; done; done
