#!/bin/bash
# round 5, session 26: record arena cap on content of 2 bytes per pixel (1024 photo_hard frames: 17.7 GB of streams).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s26
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
for CAP in 49152 36864 24576 16384; do
  KIND=photo_hard QOIMI_DEC_REC_CAP_MB=$CAP timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/photo_hard cap_mb=$CAP /"
done | tee "$OUT/dec_cap_hard.txt"
for CAP in 49152 24576; do
  KIND=sprite_alpha QOIMI_DEC_REC_CAP_MB=$CAP timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/sprite_alpha cap_mb=$CAP /"
done | tee -a "$OUT/dec_cap_hard.txt"
echo "== done"
