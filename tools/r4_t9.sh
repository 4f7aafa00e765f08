#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for m in "1.0 1.15" "1.15 1.0"; do
  set -- $m
  DRAM_W2D_MARGIN_BIG=$1 DRAM_W2D_MARGIN=$2 timeout -k 10 500 python -m pytest "tests/test_network_gpu.py::test_full_size_train_step_vs_oracle[3]" -x -q -m gpu -s > $O/t9_m$1_$2.log 2>&1; echo "margin big=$1 small=$2 rc=$?"; grep -E "max-rel|^\[config|passed|failed" $O/t9_m$1_$2.log | cut -c1-220
done
