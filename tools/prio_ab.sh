#!/bin/bash
# GPU box: A/B of the side stream's HIP priority (config 1 and config 3)
python - <<'PY'
import torch
for p in (-2, -1, 0, 1, 2):
    try:
        s = torch.cuda.Stream(priority=p); print("priority", p, "->", s.priority)
    except Exception as e:
        print("priority", p, "ERR", str(e)[:60])
PY
for p in 0 -1 1; do
  for c in 1 3; do
    DRAM_SIDE_PRIORITY=$p python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --timeline off 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('prio $p config $c', round(j['value'],2), round(j['ms_per_step'],2))"
  done
done
