#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r4; mkdir -p $O; cd $R
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
python bench.py > $O/b32_default.json 2> $O/b32_default.err; echo "rc=$?"; wc -l $O/b32_default.json
python - <<PY
import json
d=json.load(open("$O/b32_default.json"))
print({k:d[k] for k in ('metric','value','unit','n_gpus','steps','warmup','ms_per_step','higher_is_better','scaling','vs_baseline','dtype','data')})
print(d['config'])
r=d['roofline']; print({k:r[k] for k in ('bound','achieved','peak','unit','frac','traffic') if k in r})
print(d['cpu_baseline'])
print('host', d['host_issue_ms_per_step'])
PY
