for r in 1 2; do for v in "DRAM_RP_SINGLE=64" "DRAM_RP_SINGLE=1024"; do for c in 2 1; do env $v python bench.py --no-cpu-baseline --timeline off --config $c 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('$v config$c', round(d['value'],2), round(d['ms_per_step'],3))"; done; done; done
