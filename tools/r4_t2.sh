#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 300 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "streaming or winograd_path or conv1x1" > $O/t2_kernels.log 2>&1; echo "kernels rc=$?"; tail -3 $O/t2_kernels.log
export DRAM_TUNING=1
for s in 0 1; do
  DRAM_NN_STREAM=$s python bench.py --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b2_c1_stream$s.json 2>$O/b2_c1_stream$s.err
  python -c "import json;d=json.load(open('$O/b2_c1_stream$s.json'));print('config 1 nn_stream=$s', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms')"
done
for c in "1 f32" "3 f32" "2 bf16" "3 bf16"; do
  set -- $c
  for gs in 1 2; do
    DRAM_GRAPH_STREAMS=$gs python bench.py --config $1 --dtype $2 --graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b2_c$1_$2_gs$gs.json 2>$O/b2_c$1_$2_gs$gs.err
    python -c "import json;d=json.load(open('$O/b2_c$1_$2_gs$gs.json'));print('config $1 $2 graph streams=$gs', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', 'graph' if d['config']['hip_graph'] else 'EAGER(capture failed)', round(d['peak_hbm_gb'],1),'GB')"
    tail -2 $O/b2_c$1_$2_gs$gs.err | cut -c1-300
  done
done
timeout -k 10 1000 python -m pytest tests/test_network_gpu.py -x -q -m gpu -k "not golden" -s > $O/t2_net.log 2>&1; echo "net rc=$?"; grep "^\[" $O/t2_net.log | cut -c1-250; tail -3 $O/t2_net.log
