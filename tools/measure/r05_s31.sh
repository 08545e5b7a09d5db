#!/bin/bash
# round 5, session 31: 1024 sprite frames, wall clock of asynchronous encode calls beside the kernel sum (bench.py's class line read 16.65 ms, two
# 512-frame kernel sums 15.6).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s31
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
for A in 1 0; do KIND=sprite_alpha QOIMI_ENC_ALL_G2=$A timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/sprite_alpha 1024 all_g2=$A /"; done | tee "$OUT/enc_sprite_1024.txt"
KIND=photo_hard timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/photo_hard 1024 /" | tee -a "$OUT/enc_sprite_1024.txt"
echo "== done"
