#!/bin/bash
# round 6, session 44: two transcoder lanes per segment up to 256 / 512 bytes (QOIMI_DEC_SPLIT_MAX): lone frames that take those sizes
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s44
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for M in 128 256 512 1024; do
  for C in "3840 2160 photo_hard" "3840 2160 noise" "5120 2880 photo" "8192 8192 photo" "3840 2160 sprite_alpha"; do set -- $C
    QOIMI_DEC_SPLIT_MAX=$M W=$1 H=$2 KIND=$3 STATS=1 timeout 120 python tools/measure/single_trace.py 30 dec 2>&1 | tail -3 | tr '\n' ' ' | sed "s/^/split_max=$M $1x$2 $3: /"; echo
  done
done | sed -E "s/encode [0-9.]+ us, //; s/'rounds': 1, 'redo_segments': 0, //; s/'dec_chain_slots.*$//" | tee "$OUT/split_max.txt"
