#!/bin/bash
# GPU box: where does conv3_bf16_kernel's time go?  Rebuilds the library with one feature removed at a time
# (DRAM_BF16_ABL, csrc/conv_bf16.hip: 1 no weight DMA, 2 no halo DMA, 3 no LDS operand reads, 4 no MFMAs, 5 no stores,
# 6 no barriers, 7 weight DMA issued but never waited for, 8 weight DMA always from the same 12 KB) and times config 2's
# layers on the 4-wave 256-voxel kernel.   gpurun -- 'bash tools/conv_bf16_ablate.sh'
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export DRAM_TUNING=1   # ablation builds + A/B switches count under DRAM_TUNING=1 only
export DRAM_BF16_NW=4
for a in ${ABLS:-0 1 2 3 4 5 6 7 8}; do
  export DRAM_EXTRA_HIPCC_FLAGS="-DDRAM_BF16_ABL=$a"
  echo "== ABL=$a"
  python tools/conv_bf16_bench.py 3 2>&1 | grep -E "layer1 |layer4 |us2.1|us2.0|per step"
done
