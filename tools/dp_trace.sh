#!/bin/bash
# rocprofv3 kernel trace of the data-parallel step with its collectives in place at world size 1
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/dp; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/trace_fd
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_fd -- python3 $R/bench.py --config 1 --force-dist --steps 4 --warmup 2 --no-cpu-baseline --timeline off > $O/prof_fd.json 2>$O/prof_fd.err
find $O/trace_fd -name "*agent_info.csv" -delete
python3 - <<PY
import csv,glob
f=glob.glob("$O/trace_fd/*/*kernel_trace.csv")[0]
rows=list(csv.DictReader(open(f)))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
idx=[i for i,r in enumerate(rows) if 'adam_multi' in r['Kernel_Name']]
a,b=idx[-2],idx[-1]
step=rows[a+1:b+1]
t0=int(step[0]['Start_Timestamp']); t1=int(step[-1]['End_Timestamp'])
print("step span ms", (t1-t0)/1e6, "kernels", len(step))
# busy union
iv=sorted((int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in step)
busy=0; cs,ce=iv[0]
for s,e in iv[1:]:
    if s>ce: busy+=ce-cs; cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print("busy union ms", busy/1e6, "idle ms", (t1-t0-busy)/1e6)
import collections
d=collections.defaultdict(lambda:[0,0])
for r in step:
    n=r['Kernel_Name']
    if 'ccl' in n.lower() or 'nccl' in n.lower() or 'rccl' in n.lower():
        d[n[:80]][0]+=1; d[n[:80]][1]+=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
for k,v in d.items(): print(v[0], round(v[1]/1e3,1),"us total", k)
# gaps > 15 us: what precedes/follows
prev_end=None; gaps=[]
cs,ce=iv[0]
ends=sorted(step,key=lambda r:int(r['Start_Timestamp']))
cur_end=int(ends[0]['End_Timestamp']); last=ends[0]
for r in ends[1:]:
    s=int(r['Start_Timestamp'])
    if s-cur_end>15000: gaps.append(((s-cur_end)/1e3,last['Kernel_Name'][:50],r['Kernel_Name'][:50]))
    if int(r['End_Timestamp'])>cur_end: cur_end=int(r['End_Timestamp']); last=r
print("gaps >15us:", len(gaps), "sum ms", sum(g[0] for g in gaps)/1e3)
for g in sorted(gaps,reverse=True)[:25]: print(round(g[0],1), '|', g[1], '->', g[2])
PY
