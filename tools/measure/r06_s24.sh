#!/bin/bash
# round 6, session 24: the mixed directory's decode, launch by launch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${SESSION:-r06_s24}
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
timeout 300 python tools/measure/mixed_trace.py 2>&1 | tail -3 | tee "$OUT/mixed_plain.txt"
rm -rf /tmp/mt; timeout 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/mt -o t -- python tools/measure/mixed_trace.py > /tmp/mt.log 2>&1
python - <<'PY' | tee "$OUT/mixed_timeline.txt"
import csv, glob
rows = []
for f in glob.glob("/tmp/mt/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").split("(")[0]))
rows.sort()
# the last decode call that is not under profiling: find calls = sequences starting at dec_transcode<0 ...> up to dec_fill; print the one before the last two
starts = [i for i, r in enumerate(rows) if "dec_transcode<0" in r[2]]
print(len(rows), "launches,", len(starts), "decode calls")
i0 = starts[-5]; i1 = starts[-3]
t0 = rows[i0][0]; prev = t0
for s, e, n in rows[i0:i1]:
    if "enc_" in n or "synth" in n or "hash" in n: break
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:9.1f} {(s - prev) / 1e3:8.1f}  {n[:100]}")
    prev = e
PY
