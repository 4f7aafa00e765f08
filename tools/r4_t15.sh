#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
export DRAM_TUNING=1
for v in 1 3 2; do
  echo "== 64->64 @2x32x64x64 fused, DRAM_W2D_V=$v"
  DRAM_W2D_V=$v DRAM_CONV_ALGO=3 python tools/conv_bench.py 2 32 64 64 64 64 3 1 1 fwd,dgrad 20 2>&1 | grep -E "^fwd|^dgrad" | cut -c1-8,88-130
done
for v in 1 3 2; do
  echo "== 64->32 @2x64x128x128 fused, DRAM_W2D_V=$v"
  DRAM_W2D_V=$v DRAM_CONV_ALGO=3 python tools/conv_bench.py 2 64 128 128 64 32 3 1 1 fwd,dgrad 10 2>&1 | grep -E "^fwd|^dgrad" | cut -c1-8,88-130
done
