#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
python tools/host_profile.py 1 f32 5 > $O/hostprof_c1.txt 2>&1
python tools/host_profile.py 3 bf16 5 > $O/hostprof_c3bf.txt 2>&1
head -50 $O/hostprof_c1.txt
