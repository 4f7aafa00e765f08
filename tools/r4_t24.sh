#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
for c in 1 3; do
python bench.py --config $c --force-dist --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b24_fd$c.json 2>$O/b24_fd$c.err
python -c "import json;d=json.load(open('$O/b24_fd$c.json'));print('config $c force-dist', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', d.get('exposed_collective_ms'), d.get('collectives_per_step'))"
done
python bench.py --config 2 --dtype bf16 --force-dist --steps 10 --warmup 3 --no-cpu-baseline --timeline off > $O/b24_fd2.json 2>$O/b24_fd2.err
python -c "import json;d=json.load(open('$O/b24_fd2.json'));print('config 2 bf16 force-dist', round(d['value'],2),'vol/s', round(d['ms_per_step'],3),'ms', d.get('exposed_collective_ms'))"
