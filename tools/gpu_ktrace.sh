cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/kt; mkdir -p $OUT; export TMPDIR=/tmp
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OLDPWD/$OUT/run -o kt -- python $OLDPWD/bench.py --steps 1 --warmup 0 --no-cpu --kind uiflat --frames 32) > $OUT/run.log 2>&1
python - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/run/**/*kernel_trace.csv',recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if 'dec_' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
for r in rows[-70:]:
    print(r['Kernel_Name'].split('(')[0][-28:], (int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000, r['Grid_Size_X'] if 'Grid_Size_X' in r else r.get('Grid_Size'))
PY
