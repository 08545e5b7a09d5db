#!/bin/bash
# round 6, session 22: segments below 128 bytes on the single-pass path (two transcoder lanes per segment): 4K and smaller frames by segment size
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s22
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for S in "3840 2160" "2560 1440" "1920 1080" "1280 720"; do set -- $S; for K in photo sprite_alpha; do for B in "" 64 80 96 112 128; do
  W=$1 H=$2 KIND=$K QOIMI_SEG_BYTES=$B timeout 120 python tools/measure/single_trace.py 40 dec 2>&1 | tail -1 | sed "s/^/$1x$2 $K B=${B:-auto} /"
done; done; done | tee "$OUT/single_small_seg.txt"
