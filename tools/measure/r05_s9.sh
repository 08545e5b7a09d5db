#!/bin/bash
# round 5, session 9: the 16-bit-record build with P3's first table read-ahead taken from the expanded record; refinement passes on flat content.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s9
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
R16="$PWD/build/rec16/libqoi_mi355x.so"
echo "== pytest (decode tests) against the 16-bit-record build"
QOIMI_LIB=$R16 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "decode or 4k_frame or batch or hostile or selectable or record or flat" > "$OUT/pytest_rec16.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_rec16.log"; tail -4 "$OUT/pytest_rec16.log"; rm -f gpucore.* core.*
echo "== decode kernels: 32-bit records / 16-bit records"
for K in photo photo_hard; do
  KIND=$K timeout 300 python tools/measure/dec_time.py - 256 2>&1 | tail -1 | sed "s/^/$K 256 rec32 /"
  KIND=$K timeout 300 python tools/measure/dec_time.py $R16 256 2>&1 | tail -1 | sed "s/^/$K 256 rec16 /"
done | tee "$OUT/dec_rec16.txt"
KIND=photo timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/photo 1024 rec32 /" | tee -a "$OUT/dec_rec16.txt"
KIND=photo timeout 300 python tools/measure/dec_time.py $R16 1024 2>&1 | tail -1 | sed "s/^/photo 1024 rec16 /" | tee -a "$OUT/dec_rec16.txt"
echo "== PMC: HBM traffic of the 16-bit build's decode kernels (1024 photographs)"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && KIND=photo REPS=2 timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OLDPWD/$OUT/pmc_$set" -o pmc -- python "$OLDPWD/tools/measure/dec_time.py" $R16 1024) > "$OUT/pmc_$set.log" 2>&1
  echo "$set rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(list)
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'qoimi::dec' in k:
            rows[k].append((int(r['Grid_Size']), r['Counter_Name'], float(r['Counter_Value'])))
kern = {}
for k in sorted(rows):
    g = max(x[0] for x in rows[k]); e = {'grid': g}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        v = [x[2] for x in rows[k] if x[0] == g and x[1] == c]
        if v: e[c + '_KB'] = round(sum(v) / len(v), 1)
    kern[k] = e
json.dump({"build": "build/rec16 (-DQOIMI_REC16=1)", "frames": 1024, "read_correction": 2.0, "kernels": kern}, open(out + '/pmc_traffic_rec16.json', 'w'), indent=1)
for k, v in kern.items():
    if v.get('FETCH_SIZE_KB', 0) > 1e5: print(k[:50], round(v['FETCH_SIZE_KB'] * 2 * 1024 / 1e9, 2), 'GB read', round(v.get('WRITE_SIZE_KB', 0) * 1024 / 1e9, 2), 'GB written')
PY
echo "== refinement passes per round on flat content (1024 uiflat frames)"
for I in 4 8 16; do KIND=uiflat QOIMI_DEC_INNER=$I timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/uiflat inner=$I /"; done | tee "$OUT/dec_uiflat_inner.txt"
echo "== done"
