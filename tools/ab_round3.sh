for c in 1 2; do for g in "" "--graph"; do python bench.py --no-cpu-baseline --timeline off --config $c $g 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('config$c $g', round(d['value'],2), round(d['ms_per_step'],3), d['config'].get('hip_graph'))"; done; done
