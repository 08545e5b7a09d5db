#!/bin/bash
# kernel timeline of the last step of a bench run (diagnostic): start offset, duration, gap to the previous kernel
# usage: ARGS="--frames 1 --kind photo" bash tools/gpu_ktrace.sh [outdir-name]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-kt}; mkdir -p $OUT; export TMPDIR=/tmp
ARGS=${ARGS:---frames 1}
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$OUT/run -o kt -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu --no-others --no-single $ARGS) > $OUT/run.log 2>&1
python - $OUT <<'PY' | tee $OUT/timeline.txt
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/run/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'qoimi' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step = from the last enc_slab_summary launch on
idx=[i for i,r in enumerate(rows) if 'enc_slab_summary' in r['Kernel_Name']]
rows=rows[idx[-1]:] if idx else rows[-60:]
t0=int(rows[0]['Start_Timestamp']); prev_end=t0; busy=0
for r in rows:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].split('(')[0].replace('qoimi::','').replace('void ','')[:44]
    print(f"{(s-t0)/1000:9.1f} us  dur {(e-s)/1000:8.1f}  gap {(s-prev_end)/1000:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size'))}  {name}")
    busy+=e-s; prev_end=e
print(f"total {(prev_end-t0)/1000:.1f} us, kernels busy {busy/1000:.1f} us, {len(rows)} launches")
PY
