#!/bin/bash
# round 6, session 30: batches of 8 .. 256 4K frames by segment size and content class (what the cost model of choose_seg_bytes should say)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s30
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for K in photo sprite_alpha photo_hard; do for F in 8 32 128 256; do for B in "" 256 512 1024 2048 4096; do
  KIND=$K QOIMI_SEG_BYTES=$B timeout 200 python tools/measure/dec_time.py - $F 2>&1 | tail -1 | sed -E "s/^.*'rounds'/rounds/" | sed "s/^/$K F=$F B=${B:-auto} /"
done; done; done | tee "$OUT/batch_by_seg.txt"
