#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
echo "== default"; python tools/conv_bf16_bench.py 20 2>&1 | grep -v amdgpu.ids
echo "== ZWALK_WGS=256"; DRAM_BF16_ZWALK_WGS=256 python tools/conv_bf16_bench.py 20 2>&1 | grep -v amdgpu.ids | grep "layer1\|us1.1\|us2\|us3\|per step"
echo "== ZWALK_WGS=1024"; DRAM_BF16_ZWALK_WGS=1024 python tools/conv_bf16_bench.py 20 2>&1 | grep -v amdgpu.ids | grep "layer1\|us1.1\|us2\|us3\|per step"
echo "== WGRAD=tile"; DRAM_BF16_WGRAD=tile python tools/conv_bf16_bench.py 20 2>&1 | grep -v amdgpu.ids | grep "layer1\|us1.1\|us2\|us3\|per step"
