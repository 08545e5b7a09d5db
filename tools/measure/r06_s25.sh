#!/bin/bash
# round 6, session 25: the mixed directory's decode by segment size
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s25
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for B in "" 128 256 512 1024 2048 4096; do
  QOIMI_SEG_BYTES=$B timeout 300 python tools/measure/mixed_trace.py 2>&1 | tail -2 | sed "s/^/B=${B:-auto} /"
done | tee "$OUT/mixed_by_seg.txt"
