#!/bin/bash
# round 6, session 43: the first sync run-up of small segments at 16 / 24 / 32 bytes (the second one behind it): lone frames
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s43
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for R in 16 24 32; do
  L=build/exp_short$R/libqoi_mi355x.so; [ $R = 32 ] && L=qoi_amd/lib/libqoi_mi355x.so
  for S in "3840 2160" "1920 1080" "1280 720"; do set -- $S; for K in photo photo_hard sprite_alpha; do
    QOIMI_TOOLS_LIB=$L W=$1 H=$2 KIND=$K STATS=1 timeout 120 python tools/measure/single_trace.py 40 dec 2>&1 | tail -3 | tr '\n' ' ' | sed "s/^/short=$R $1x$2 $K: /"; echo
  done; done
done | sed -E "s/encode [0-9.]+ us, //; s/'rounds': 1, 'redo_segments': 0, //; s/'dec_chain_slots.*$//" | tee "$OUT/short_len.txt"
