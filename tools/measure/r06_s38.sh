#!/bin/bash
# round 6, session 38: a second, longer sync run-up in dec_transcode<0> for lanes whose chains did not meet: lone frames by class, mixed directory, batches, the GPU suite
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s38
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for S in "3840 2160" "1280 720"; do set -- $S; for K in photo sprite_alpha photo_hard noise; do
  W=$1 H=$2 KIND=$K STATS=1 timeout 120 python tools/measure/single_trace.py 40 dec 2>&1 | tail -3 | sed "s/^/$1x$2 $K: /"
done; done | tee "$OUT/single_kinds.txt"
timeout 300 python tools/measure/mixed_trace.py 2>&1 | tail -2 | cut -c1-1200 | tee "$OUT/mixed.txt"
for K in photo sprite_alpha photo_hard; do for F in 32 256; do
  KIND=$K timeout 200 python tools/measure/dec_time.py - $F 2>&1 | tail -1 | sed "s/^/$K F=$F /" | cut -c1-400
done; done | tee "$OUT/batch_auto.txt"
unset QOIMI_TUNING
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee "$OUT/pytest.txt"
