#!/bin/bash
# round 6, session 10: what makes 256-byte segments pathological for flat frames (dec_segments 142 ms per 128 frames)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s10
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for B in 128 256 384 640; do for RD in 2 0; do
  KIND=uiflat QOIMI_SEG_BYTES=$B QOIMI_DEC_RUN_DESC=$RD timeout 200 python tools/measure/dec_time.py - 128 2>&1 | tail -1 | cut -c1-330 | sed "s/^/uiflat B=$B run_desc=$RD /"
done; done | tee "$OUT/flat_256.txt"
KIND=photo QOIMI_SEG_BYTES=256 timeout 200 python tools/measure/dec_time.py - 128 2>&1 | tail -1 | cut -c1-330 | sed "s/^/photo B=256 /" | tee -a "$OUT/flat_256.txt"
KIND=constant QOIMI_SEG_BYTES=256 timeout 200 python tools/measure/dec_time.py - 128 2>&1 | tail -1 | cut -c1-330 | sed "s/^/constant B=256 /" | tee -a "$OUT/flat_256.txt"
