#!/bin/bash
# the whole GPU suite, one process, progress into gpurun_out/suite/full.log
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/suite; mkdir -p $O; cd $R
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu --durations=25 > $O/full.log 2>&1; rc=$?
tail -45 $O/full.log
exit $rc
