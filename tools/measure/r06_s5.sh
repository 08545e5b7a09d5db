#!/bin/bash
# round 6, session 5: host side of a lone frame's decode: where the copies sit between the kernels (kernel + memory-copy trace)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s5
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
R=$PWD
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d "$R/$OUT/trace" -o t -- python "$R/tools/measure/single_trace.py" 30 dec) > "$OUT/trace.log" 2>&1
ls "$OUT/trace"/* | head
python - "$OUT/trace" <<'PY' | tee "$OUT/host_timeline.txt"
import csv, glob, sys
d = sys.argv[1]
ev = []
for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K " + r["Kernel_Name"].replace("void ", "").split("(")[0][:50]))
for f in glob.glob(d + "/**/*memory_copy_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "?")))
for f in glob.glob(d + "/**/*hip_api_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "A " + r["Function"]))
ev.sort()
# the last complete decode call: from the last-but-one H2D copy on
idx = [i for i, e in enumerate(ev) if e[2].startswith("K qoimi::dec_transcode<0>")]
lo = idx[-3]; hi = idx[-2]
t0 = ev[lo][0]
for s, e, n in ev[max(0, lo - 12):hi]:
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n}")
PY
rm -rf "$OUT/trace"
