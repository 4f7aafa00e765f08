set -e
mkdir -p gpurun_out/ab
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -x -q -m gpu > gpurun_out/ab/pytest.txt 2>&1 || { tail -40 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
for v in "DRAM_BF16_S2=0" "DRAM_BF16_S2=1"; do for c in 2; do env $v python bench.py --no-cpu-baseline --timeline off --config $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v config$c', round(d['value'],2), round(d['ms_per_step'],3))"; done; done
python bench.py --no-cpu-baseline --timeline off --config 3 --dtype bf16 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config3 bf16', round(d['value'],2), round(d['ms_per_step'],3))"
