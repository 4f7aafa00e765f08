#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 1000 python -m pytest tests/test_network_gpu.py -x -q -m gpu -s -k "not golden and not mid_size" --deselect "tests/test_network_gpu.py::test_full_size_train_step_vs_oracle[1]" --deselect "tests/test_network_gpu.py::test_full_size_train_step_vs_oracle[2]" > $O/t10_net.log 2>&1; echo "net rc=$?"; grep -E "^\[config|passed|failed|max-rel" $O/t10_net.log | cut -c1-220
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "winograd_path" > $O/t10_k.log 2>&1; echo "k rc=$?"; tail -2 $O/t10_k.log
for c in "1 f32" "3 f32" "5 f32" "0 f32"; do
  set -- $c
  python bench.py --config $1 --dtype $2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b10_c$1_$2.json 2>$O/b10_c$1_$2.err
  python -c "import json;d=json.load(open('$O/b10_c$1_$2.json'));print('config $1 $2', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'eager', round(d['peak_hbm_gb'],1),'GB')"
done
