#!/bin/bash
# round 5, session 27: wall clock of whole decode calls against the sum of their kernels, by record arena cap (photo_hard, photo).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s27
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
for K in photo_hard photo; do for CAP in 49152 24576; do
  KIND=$K QOIMI_DEC_REC_CAP_MB=$CAP timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K cap_mb=$CAP /"
done; done | tee "$OUT/dec_cap_wall.txt"
echo "== done"
