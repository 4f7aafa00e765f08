set -e
timeout -k 10 400 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "inplane" 2>&1 | tail -12
export DRAM_CONV_ALGO=3
for v in 1 2; do
echo "== DRAM_W2D_V=$v"
export DRAM_W2D_V=$v
timeout -k 10 120 python tools/conv_bench.py 2 64 128 128 64 64 3 1 1 fwd,dgrad 2>&1 | grep TFLOP
timeout -k 10 120 python tools/conv_bench.py 2 64 128 128 128 64 3 1 1 fwd,dgrad 2>&1 | grep TFLOP
timeout -k 10 120 python tools/conv_bench.py 2 32 64 64 64 64 3 1 1 fwd,dgrad 2>&1 | grep TFLOP
timeout -k 10 120 python tools/conv_bench.py 2 64 128 128 64 32 3 1 1 fwd,dgrad 2>&1 | grep TFLOP
timeout -k 10 120 python tools/conv_bench.py 2 32 64 64 576 64 3 1 1 fwd,dgrad 2>&1 | grep TFLOP
done
