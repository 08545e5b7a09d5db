#!/bin/bash
# round 6, session 3: what dec_scan_entry's 20 us are made of (timing-only builds: tail walk / look-back / ticket compiled out)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s3
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
R=$PWD
for L in qoi_amd/lib build/exp_scan1 build/exp_scan2 build/exp_scan4 build/exp_scan7; do
  (cd /tmp && QOIMI_TOOLS_LIB=$R/$L/libqoi_mi355x.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/trace" -o t -- python "$R/tools/measure/single_trace.py" 60 dec) > "$OUT/trace.log" 2>&1
  python tools/measure/trace_timeline.py "$OUT/trace" "dec_transcode<0" 40 | grep -E "dec_scan_entry|dec_transcode" | sed "s|^|$L |"
  rm -rf "$OUT/trace"
done | tee "$OUT/scan_parts.txt"
