#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py tests/test_bf16_gpu.py -x -q -m gpu -k "batchnorm or bn_ or elementwise" > $O/t14_k.log 2>&1; echo "k rc=$?"; tail -2 $O/t14_k.log
bash tools/r4_prof.sh > /dev/null 2>&1
cd $R
for c in "3 bf16" "1 f32"; do set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b14_c$1.json 2>/dev/null
  python -c "import json;d=json.load(open('$O/b14_c$1.json'));print('config $1 $2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
