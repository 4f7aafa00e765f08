#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
export DRAM_TUNING=1
for spec in "2 64 128 128 64 64 3 1 1" "2 32 64 64 64 64 3 1 1" "2 64 128 128 128 64 3 1 1"; do
  for algo in 2 3; do
    echo "== $spec DRAM_CONV_ALGO=$algo"
    DRAM_CONV_ALGO=$algo python tools/conv_bench.py $spec fwd,dgrad,wgrad 10 2>&1 | grep -E "plan|fwd|dgrad|wgrad|cached"
  done
done
