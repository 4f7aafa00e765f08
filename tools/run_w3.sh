timeout -k 10 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "winograd_path" 2>&1 | tail -2
export DRAM_CONV_ALGO=2
timeout -k 10 120 python tools/conv_bench.py 2 16 32 32 512 512 3 1 4 wgrad 2>&1 | grep TFLOP | awk '{print $1, $(NF-4), $(NF-3), $(NF-1)}'
timeout -k 10 120 python tools/conv_bench.py 2 16 32 32 256 256 3 1 2 wgrad 2>&1 | grep TFLOP | awk '{print $1, $(NF-4), $(NF-3), $(NF-1)}'
timeout -k 10 120 python tools/conv_bench.py 2 16 32 32 256 512 3 1 4 wgrad 2>&1 | grep TFLOP | awk '{print $1, $(NF-4), $(NF-3), $(NF-1)}'
