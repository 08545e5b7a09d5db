#!/bin/bash
# round 6, session 42: the second sync run-up by length (builds build/exp_retry<N>): lone frames by class - time and segments still flagged
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s42
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for R in 64 96 128 192; do
  L=build/exp_retry$R/libqoi_mi355x.so; [ $R = 192 ] && L=qoi_amd/lib/libqoi_mi355x.so
  for S in "3840 2160" "1280 720"; do set -- $S; for K in sprite_alpha photo_hard noise; do
    QOIMI_TOOLS_LIB=$L W=$1 H=$2 KIND=$K STATS=1 timeout 120 python tools/measure/single_trace.py 40 dec 2>&1 | tail -3 | head -2 | tr '\n' ' ' | sed "s/^/retry=$R $1x$2 $K: /"; echo
  done; done
done | sed -E "s/encode [0-9.]+ us, //; s/'rounds': 1, 'redo_segments': 0, //" | tee "$OUT/retry_len.txt"
