#!/bin/bash
# busy cycles (GRBM_GUI_ACTIVE) and duration of enc_sets for two builds: tools/measure/clock_ab.sh libA libB
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/clk_ab; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
i=0
for lib in "$@"; do
  i=$((i+1))
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $OLDPWD/$OUT/arm$i -o pmc -- python $OLDPWD/tools/measure/enc_time.py $OLDPWD/$lib 256) > $OUT/arm$i.log 2>&1; echo "$lib rc=$?"
done
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for arm in sorted(glob.glob(out + '/arm*/')):
    cnt = {}; dur = {}
    for f in glob.glob(arm + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'enc_sets<4, 1, 1>' in r['Kernel_Name'] and r['Counter_Name'] == 'GRBM_GUI_ACTIVE': cnt[int(r['Dispatch_Id'])] = float(r['Counter_Value'])
    for f in glob.glob(arm + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'enc_sets<4, 1, 1>' in r['Kernel_Name']: dur[int(r['Dispatch_Id'])] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    ids = sorted(set(cnt) & set(dur))[2:]
    c = sum(cnt[i] for i in ids) / len(ids); d = sum(dur[i] for i in ids) / len(ids)
    print(arm, f"launches {len(ids)} GRBM_GUI_ACTIVE {c:.4g} (per XCD {c/8:.4g}) duration {d/1e3:.1f} us -> {c/8/d*1e3:.0f} MHz")
PY
