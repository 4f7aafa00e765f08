#!/bin/bash
# per-layer single-stream tables of the four configs the round tunes (runs on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r4
mkdir -p $O
cd $R
export DRAM_TUNING=1 DRAM_WGRAD_STREAM=0
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --detail $O/base_c1_layers.txt > $O/base_c1.json 2>$O/base_c1.err
python bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --detail $O/base_c3_layers.txt > $O/base_c3.json 2>/dev/null
python bench.py --config 3 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --detail $O/base_c3bf_layers.txt > $O/base_c3bf.json 2>/dev/null
python bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --detail $O/base_c2_layers.txt > $O/base_c2.json 2>/dev/null
unset DRAM_WGRAD_STREAM
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/base_c1_plain.json 2>/dev/null
tail -c 600 $O/base_c1_plain.json
