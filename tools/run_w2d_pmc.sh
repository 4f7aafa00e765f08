set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DRAM_CONV_ALGO=3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace --output-format csv -d $R/gpurun_out/w2d_pmc -- python3 $R/tools/conv_bench.py 2 64 128 128 64 64 3 1 1 fwd 3 > $R/gpurun_out/w2d_pmc.log 2>&1
ls $R/gpurun_out/w2d_pmc/*/ | head
