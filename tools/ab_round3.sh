for h in 1 3 4 5; do echo "== DRAM_WINO_HALF=$h"; DRAM_WINO_HALF=$h python tools/conv_bench.py 2 64 128 128 128 64 3 1 1 fwd,dgrad 10 2>&1 | grep ConvGeom | cut -c1-8,90-130; DRAM_WINO_HALF=$h python tools/conv_bench.py 2 16 32 32 512 512 3 1 4 fwd 10 2>&1 | grep ConvGeom | cut -c1-8,90-130; done
DRAM_WINO_HALF=4 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wino" 2>&1 | tail -2
DRAM_WINO_HALF=5 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "wino" 2>&1 | tail -2
