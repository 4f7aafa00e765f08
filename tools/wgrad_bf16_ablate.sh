#!/bin/bash
# GPU box: where does wgrad3b_bf16_kernel's time go?  Rebuilds the library with one feature removed at a time
# (DRAM_BF16_ABL: 11 no tile DMA, 12 no LDS operand reads, 13 no MFMAs) and times the weight gradient of config 2's layers.
R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
export DRAM_TUNING=1   # ablation builds + A/B switches count under DRAM_TUNING=1 only
for a in ${ABLS:-0 11 12 13}; do
  export DRAM_EXTRA_HIPCC_FLAGS="-DDRAM_BF16_ABL=$a"
  echo "== ABL=$a"
  python tools/conv_bf16_bench.py 5 2>&1 | grep -E "layer1 |layer3 |layer4 |us1.0|per step" | sed 's/fwd.*wgrad/wgrad/'
done
