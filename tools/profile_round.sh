#!/bin/bash
# Runs ON THE GPU BOX (via gpurun): the bench lines, the rocprofv3 kernel trace and the two PMC
# passes (FETCH_SIZE / WRITE_SIZE in separate runs) behind profiles/rNN_*.  Output: gpurun_out/prof/.
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh bench'     (the bench lines)
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh prof1'     (the rocprofv3 passes of config 1)
#   gpurun --timeout 1200 -- 'bash tools/profile_round.sh prof2'     (... of config 2 as specified and of ResNet-50; `prof` = both, may exceed one call's 20 min)
# (delete the local gpurun_out/prof first: gpurun merges, it does not mirror)
# every command's stderr is kept in a file next to its output (an empty .json.log then has its cause on record)
# then locally:  python tools/summarize_profile.py gpurun_out/prof profiles r02
set -e
PHASE=${1:-all}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/prof
mkdir -p $O
cd $R
if [ "$PHASE" = "all" ] || [ "$PHASE" = "bench" ]; then
python bench.py --steps 20 --warmup 5 > $O/bench_config1.json.log 2>$O/bench_config1.err
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --detail $O/bench_config1_per_layer.txt > $O/bench_config1_detail.json.log 2>$O/bench_config1_detail.json.log.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --timeline off --no-graph > $O/bench_config1_no_timeline.json.log 2>$O/bench_config1_no_timeline.json.log.err
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --timeline off --graph > $O/bench_config1_graph.json.log 2>$O/bench_config1_graph.json.log.err
for c in 0 3 5; do python bench.py --config $c --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_config$c.json.log 2>$O/bench_config$c.json.log.err; done
python bench.py --config 1 --graph --graph-streams 2 --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/bench_config1_graph2.json.log 2>$O/bench_config1_graph2.json.log.err     # (default: one branch)
python bench.py --config 3 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/bench_config3_eager.json.log 2>$O/bench_config3_eager.json.log.err
python bench.py --config 2 --no-graph --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/bench_config2_eager.json.log 2>$O/bench_config2_eager.json.log.err
python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_config2.json.log 2>$O/bench_config2.json.log.err          # bf16, as specified
python bench.py --config 2 --dtype f32 --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_config2_f32.json.log 2>$O/bench_config2_f32.json.log.err
python bench.py --config 3 --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_config3_bf16.json.log 2>$O/bench_config3_bf16.json.log.err
python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_config4.json.log 2>$O/bench_config4.json.log.err           # bf16 + recompute
# the data-parallel step at world size 1 with every collective in place (own RCCL communicators), next to the plain step
# of the same process (plain_ms_per_step / exposed_collective_ms in each line)
for spec in "1 f32" "2 bf16" "3 bf16"; do
  set -- $spec
  python bench.py --config $1 --dtype $2 --force-dist --steps 20 --warmup 5 --no-cpu-baseline --timeline off > $O/bench_config$1_$2_forcedist.json.log 2>$O/bench_config$1_$2_forcedist.json.log.err
done
fi
if [ "$PHASE" = "bench" ]; then ls -la $O | head -60; exit 0; fi
cd /tmp && export TMPDIR=/tmp
# the profiled passes run the step on ONE stream, like bench.py's timeline pass: with the weight-gradient kernels
# overlapping the data-gradient chain on a second stream, per-kernel durations contain the time a kernel shared its
# CUs with the other stream's kernel and would not be comparable
export DRAM_TUNING=1 DRAM_WGRAD_STREAM=0
if [ "$PHASE" != "prof2" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --timeline off --no-graph > $O/bench_config1_under_rocprofv3.json.log 2>$O/bench_config1_under_rocprofv3.json.log.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timeline off --no-graph > $O/pmc_fetch.out 2>$O/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timeline off --no-graph > $O/pmc_write.out 2>$O/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --timeline off --no-graph > $O/pmc_sq.out 2>$O/pmc_sq.err
fi
if [ "$PHASE" != "prof1" ]; then
# the same four passes for BASELINE configs[2] as specified (bf16 storage path)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c2 -- python3 $R/bench.py --config 2 --steps 5 --warmup 2 --no-cpu-baseline --timeline off --no-graph > $O/bench_config2_under_rocprofv3.json.log 2>$O/bench_config2_under_rocprofv3.json.log.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch_c2 -- python3 $R/bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline --timeline off --no-graph > $O/pmc_fetch_c2.out 2>$O/pmc_fetch_c2.err
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write_c2 -- python3 $R/bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline --timeline off --no-graph > $O/pmc_write_c2.out 2>$O/pmc_write_c2.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O/pmc_sq_c2 -- python3 $R/bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline --timeline off --no-graph > $O/pmc_sq_c2.out 2>$O/pmc_sq_c2.err
# ResNet-50 (config 3), fp32 and bf16: kernel traces
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3 -- python3 $R/bench.py --config 3 --steps 5 --warmup 2 --no-cpu-baseline --timeline off --no-graph > $O/trace_c3.out 2>$O/trace_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_c3bf -- python3 $R/bench.py --config 3 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline --timeline off --no-graph > $O/trace_c3bf.out 2>$O/trace_c3bf.err
find $O/trace_c3 $O/trace_c3bf -name "*kernel_trace.csv" -delete
fi
# keep only the small csv files (kernel traces of the PMC passes are not needed)
find $O -name "*agent_info.csv" -delete
for dd in $O/pmc_fetch $O/pmc_write $O/pmc_sq $O/pmc_fetch_c2 $O/pmc_write_c2 $O/pmc_sq_c2; do
  if [ -d $dd ]; then find $dd -name "*kernel_trace.csv" -delete; fi
done
ls -la $O $O/*/* | head -40
