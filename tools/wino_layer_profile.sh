set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "2 64 128 128 128 64 3 1 1" "2 16 32 32 512 512 3 1 4" "2 32 64 64 576 64 3 1 1"; do
 for mode in fwd dgrad wgrad; do
  tag=$(echo $spec | tr ' ' '_')_$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wp/$tag -- python3 $R/tools/conv_bench.py $spec $mode 5 > $R/gpurun_out/wp/$tag.log 2>&1
  f=$(find $R/gpurun_out/wp/$tag -name '*kernel_stats.csv' | head -1)
  echo "== $tag"; tail -1 $R/gpurun_out/wp/$tag.log; head -8 $f | cut -c1-160
 done
done
