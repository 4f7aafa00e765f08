set -e
mkdir -p gpurun_out/ab
python -m pytest tests/test_kernels_gpu.py -x -q -m gpu > gpurun_out/ab/pytest.txt 2>&1 || { tail -30 gpurun_out/ab/pytest.txt; exit 1; }
tail -2 gpurun_out/ab/pytest.txt
rm -f gpurun_out/ab/wino_epi.txt
for epi in 0 1; do
  echo "== DRAM_WINO_EPI=$epi" >> gpurun_out/ab/wino_epi.txt
  for spec in "2 64 128 128 128 64 3 1 1" "2 32 64 64 576 64 3 1 1" "2 16 32 32 512 512 3 1 4" "2 16 32 32 256 256 3 1 2" "2 16 32 32 128 128 3 1 1"; do
    DRAM_WINO_EPI=$epi python tools/conv_bench.py $spec fwd,dgrad 10 2>&1 | grep ConvGeom >> gpurun_out/ab/wino_epi.txt
  done
done
cat gpurun_out/ab/wino_epi.txt
for v in "DRAM_WINO_EPI=0 DRAM_WINO_NT=0 DRAM_EW_SHAPE=0" "DRAM_WINO_EPI=1 DRAM_WINO_NT=0" "DRAM_WINO_EPI=1 DRAM_WINO_NT=3" "DRAM_WINO_EPI=1 DRAM_WINO_NT=3 X=2"; do
  echo "== $v"; env $v python bench.py --no-cpu-baseline --timeline off 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['ms_per_step'])"
done
DRAM_EW_SHAPE=2 python tools/ew_bench.py 2>&1 | grep "bn_" > gpurun_out/ab/ew_shape2b.txt
python bench.py --no-cpu-baseline --timeline off --config 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config2', d['value'], d['ms_per_step'])"
