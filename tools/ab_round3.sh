for v in "X=1" "DRAM_WINO_NT=0" "DRAM_WINO_NT=0 DRAM_WINO_EPI=0" "DRAM_WINO_NT=0 DRAM_EW_SHAPE=0" "DRAM_WINO_NT=0 DRAM_EW_SHAPE=1"; do
  echo "== $v"; for c in 0 5; do env $v python bench.py --no-cpu-baseline --timeline off --config $c --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  config$c', round(d['value'],2), round(d['ms_per_step'],2))"; done
done
