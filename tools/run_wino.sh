set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "winograd" 2>&1 | tail -15
for algo in 1 2; do
  echo "== DRAM_CONV_ALGO=$algo"
  DRAM_CONV_ALGO=$algo timeout -k 10 120 python tools/conv_bench.py 2 16 32 32 512 512 3 1 4
  DRAM_CONV_ALGO=$algo timeout -k 10 120 python tools/conv_bench.py 2 16 32 32 256 256 3 1 2
  DRAM_CONV_ALGO=$algo timeout -k 10 120 python tools/conv_bench.py 2 16 32 32 256 512 3 1 4
  DRAM_CONV_ALGO=$algo timeout -k 10 120 python tools/conv_bench.py 2 16 32 32 128 256 3 1 2
  DRAM_CONV_ALGO=$algo timeout -k 10 120 python tools/conv_bench.py 2 32 64 64 576 64 3 1 1 fwd,dgrad
  DRAM_CONV_ALGO=$algo timeout -k 10 120 python tools/conv_bench.py 2 16 32 32 128 128 3 1 1 fwd,dgrad
done
