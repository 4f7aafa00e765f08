set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py tests/test_models_gpu.py -x -q -m gpu > gpurun_out/ab/pytest.txt 2>&1 || { tail -40 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
for c in 1 2; do python bench.py --no-cpu-baseline --config $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); f=d['roofline']['families']['head_loss']; print('config$c', round(d['value'],2), round(d['ms_per_step'],3), 'head_loss', round(f['ms_per_step'],3), round(f['frac'],3))"; done
