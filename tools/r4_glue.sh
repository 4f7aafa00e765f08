#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
python tools/glue_trace.py 1 > $O/glue_c1.txt 2>&1; python tools/glue_trace.py 2 > $O/glue_c2.txt 2>&1
tail -80 $O/glue_c1.txt
