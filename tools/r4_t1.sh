#!/bin/bash
# round-4 check 1: upmix kernel tests, golden nets on both decoder paths, host gate
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "upmix or upcat" > $O/t1_kernels.log 2>&1; echo "kernels rc=$?" 
tail -5 $O/t1_kernels.log
timeout -k 10 900 python -m pytest tests/test_network_gpu.py -x -q -m gpu -k "golden and (upmix or net_0 or net_6)" > $O/t1_net.log 2>&1; echo "net rc=$?"
tail -5 $O/t1_net.log
