#!/bin/bash
# round 6, session 20: one large image (16384^2, 8192^2) and small frames by segment size on the new small-call path
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s20
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for S in "16384 16384" "8192 8192" "1920 1080" "1280 720"; do set -- $S; for B in "" 128 256 512 1024 2048 4096; do
  W=$1 H=$2 QOIMI_SEG_BYTES=$B timeout 120 python tools/measure/single_trace.py 30 dec 2>&1 | tail -1 | sed "s/^/$1x$2 B=${B:-auto} /"
done; done | tee "$OUT/single_by_seg.txt"
