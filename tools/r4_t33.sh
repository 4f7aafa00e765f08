#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
python bench.py --config 2 --dtype bf16 --no-graph --steps 5 --warmup 2 --no-cpu-baseline --detail $O/c2bf_per_layer.txt > $O/b33.json 2>/dev/null
cat $O/c2bf_per_layer.txt | cut -c1-200
