#!/bin/bash
# effective shader clock under the encode / decode kernels: GRBM_GUI_ACTIVE (cycles the GPU was busy) per kernel against its duration
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-clk}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
(cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OLDPWD/$OUT/enc -o pmc -- python $OLDPWD/tools/measure/enc_time.py - 256) > $OUT/enc.log 2>&1; echo "enc rc=$?"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE GRBM_COUNT --output-format csv -d $OLDPWD/$OUT/dec -o pmc -- python $OLDPWD/bench.py --frames 256 --steps 3 --warmup 1 --no-cpu --no-others --no-single --no-configs) > $OUT/dec.log 2>&1; echo "dec rc=$?"
python - $OUT <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for arm in ('enc', 'dec'):
    cnt = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
    for f in glob.glob(f'{out}/{arm}/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if 'qoimi::' not in k: continue
            cnt[k][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for f in glob.glob(f'{out}/{arm}/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            if 'qoimi::' in k: dur[k].append((int(r['Dispatch_Id']), (int(r['End_Timestamp']) - int(r['Start_Timestamp']))))
    for k in sorted(cnt):
        d = dict(dur[k]); best = None
        for cname, vals in cnt[k].items():
            # the longest dispatch of the kernel
            did = max(d, key=lambda x: d[x]) if d else None
            v = dict(vals).get(did)
            if v is not None and did is not None:
                print(f"{arm} {k[:60]:60s} {cname}: {v:.4g} in {d[did] / 1e3:.1f} us -> {v / d[did] * 1e3:.0f} MHz-equivalent")
PY
