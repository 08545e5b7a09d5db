#!/bin/bash
# round 6, session 9: 128 flat 4K frames (a mid-size batch of UI frames) by segment size - the alternating leg of bench.py read their decode at 97 ms
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s9
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for F in 128 32 512; do for B in "" 256 512 1024 2048 4096; do
  KIND=uiflat QOIMI_SEG_BYTES=$B timeout 200 python tools/measure/dec_time.py - $F 2>&1 | tail -1 | sed "s/^/F=$F B=${B:-auto} /"
done; done | tee "$OUT/uiflat_mid_batch.txt"
