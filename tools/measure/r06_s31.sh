#!/bin/bash
# round 6, session 31: the piece parse at two and four pieces per segment (256 / 512 bytes): parity subset, mixed directory and batches by segment size, large lone images
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s31
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
QOIMI_SEG_BYTES=256 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not selectable and not small_calls" 2>&1 | tail -3 | tee "$OUT/pytest_256.txt"
QOIMI_SEG_BYTES=512 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "not selectable and not small_calls" 2>&1 | tail -3 | tee "$OUT/pytest_512.txt"
for V in "QOIMI_SEG_BYTES=" "QOIMI_SEG_BYTES=256" "QOIMI_SEG_BYTES=512" "QOIMI_SEG_BYTES=1024"; do
  env $V timeout 300 python tools/measure/mixed_trace.py 2>&1 | tail -2 | sed "s/^/$V /"
done | cut -c1-1300 | tee "$OUT/mixed_by_seg.txt"
for K in photo sprite_alpha; do for F in 8 32 128; do for B in "" 256 512 1024 2048; do
  KIND=$K QOIMI_SEG_BYTES=$B timeout 200 python tools/measure/dec_time.py - $F 2>&1 | tail -1 | sed -E "s/^.*'decode_total'/total/" | sed "s/^/$K F=$F B=${B:-auto} /"
done; done; done | tee "$OUT/batch_by_seg.txt"
for S in "16384 16384" "8192 8192" "5120 2880"; do set -- $S; for B in "" 128 256 512 1024 2048; do
  W=$1 H=$2 QOIMI_SEG_BYTES=$B timeout 120 python tools/measure/single_trace.py 30 dec 2>&1 | tail -1 | sed "s/^/$1x$2 B=${B:-auto} /"
done; done | tee "$OUT/single_by_seg.txt"
