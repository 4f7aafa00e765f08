for algo in 2 3; do
echo "== DRAM_CONV_ALGO=$algo"
export DRAM_CONV_ALGO=$algo
for shape in "2 64 128 128 128 64" "2 64 128 128 64 64" "2 64 128 128 64 32" "2 32 64 64 576 64" "2 32 64 64 64 64"; do
timeout -k 10 120 python tools/conv_bench.py $shape 3 1 1 2>&1 | grep TFLOP | awk '{print $1, $3,$4,$5,$6,$7,$8, $(NF-4), $(NF-3), $(NF-1)}'
done
done
