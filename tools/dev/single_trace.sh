#!/bin/bash
# rocprofv3 kernel trace of lone 4K frame decodes: per-kernel durations and the gaps between kernels of one call
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-strace}; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$OUT -o t -- python $OLDPWD/tools/dev/single_prof.py ${2:-photo}) > $OUT/log.txt 2>&1
tail -25 $OUT/log.txt | grep -v "^[EW]2026"
python - $OUT <<'PY'
import csv,glob,sys,collections
f=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# take the last 400 kernels that belong to qoimi decode; split into calls at dec_transcode<0>
ev=[(r['Kernel_Name'].split('(')[0].replace('void ','').replace('qoimi::',''), int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'qoimi' in r['Kernel_Name']]
calls=[];cur=[]
for e in ev:
    if e[0].startswith('dec_transcode<0>'):
        if cur: calls.append(cur)
        cur=[]
    if e[0].startswith('dec_') : cur.append(e)
calls=[c for c in calls if len(c)>10][-15:]
agg=collections.OrderedDict()
span=[]
for c in calls:
    span.append((c[-1][2]-c[0][1])/1e3)
    prev_end=None
    for i,(n,s,e) in enumerate(c):
        key=f"{i:02d} {n}"
        d=agg.setdefault(key,[0,0,0]); d[0]+=(e-s)/1e3; d[1]+=((s-prev_end)/1e3 if prev_end else 0); d[2]+=1
        prev_end=e
print('decode span (first kernel start .. last kernel end) us:', sum(span)/len(span))
for k,(dur,gap,n) in agg.items(): print(f"{k:40s} dur {dur/n:7.1f}  gap_before {gap/n:6.1f}")
PY
