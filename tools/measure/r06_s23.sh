#!/bin/bash
# round 6, session 23: lone frames of the kinds that are not photographs: rounds, re-opened segments and per-kernel times
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${SESSION:-r06_s23}
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for S in "3840 2160" "1280 720"; do set -- $S; for K in photo sprite_alpha photo_hard noise uiflat constant; do
  W=$1 H=$2 KIND=$K STATS=1 timeout 120 python tools/measure/single_trace.py 40 dec 2>&1 | tail -3 | sed "s/^/$1x$2 $K: /"
done; done | tee "$OUT/single_kinds.txt"
