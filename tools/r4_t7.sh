#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
export DRAM_TUNING=1
for spec in "2 64 128 128 128 64 3 1 1" "2 64 128 128 64 64 3 1 1" "2 32 64 64 64 64 3 1 1" "2 16 32 32 512 512 3 1 4"; do
  for pf in 0 1; do
    echo "== $spec pipeline, DRAM_WINO_PF=$pf"
    DRAM_WINO_PF=$pf DRAM_CONV_ALGO=2 python tools/conv_bench.py $spec fwd,dgrad 10 2>&1 | grep -E "^fwd|^dgrad" | cut -c1-8,90-130
  done
done
cd /tmp && export TMPDIR=/tmp
for pf in 0 1; do
DRAM_WINO_PF=$pf DRAM_CONV_ALGO=2 rocprofv3 --kernel-trace --stats --output-format csv -d $O/wp_pf$pf -- python3 $R/tools/conv_bench.py 2 64 128 128 128 64 3 1 1 fwd 5 > /dev/null 2>&1
f=$(find $O/wp_pf$pf -name '*kernel_stats.csv' | head -1); echo "pf=$pf"; head -4 $f | cut -c1-60,140-260
done
