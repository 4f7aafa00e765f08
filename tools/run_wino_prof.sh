set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export DRAM_CONV_ALGO=2
for shape in "2 16 32 32 512 512 3 1 4" "2 16 32 32 256 256 3 1 2" "2 64 128 128 128 64 3 1 1" "2 32 64 64 64 64 3 1 1"; do
  tag=$(echo $shape | tr ' ' '_')
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/wprof_$tag -- python3 $R/tools/conv_bench.py $shape fwd,dgrad,wgrad 10 > $R/gpurun_out/wprof_$tag.log 2>&1
  f=$(ls $R/gpurun_out/wprof_$tag/*/*kernel_stats.csv | head -1)
  echo "== $shape"; head -12 $f | cut -d, -f1-6 | cut -c1-150
done
