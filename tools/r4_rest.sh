#!/bin/bash
# the GPU suite from test_distributed_gpu on (after a -x stop), one process per file group
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests/test_distributed_gpu.py tests/test_kernels_gpu.py tests/test_models_gpu.py tests/test_network_gpu.py -x -q -m gpu --durations=15 > $O/rest.log 2>&1; rc=$?
tail -30 $O/rest.log
exit $rc
