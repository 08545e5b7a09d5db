#!/bin/bash
# round 5, session 5: the state look-back (ENTRY 2) with its exchanges' result register kept live; parity, then flat content and a lone frame.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s5
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== pytest: encode tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "encode or flat or fuzz or selectable or sweep or 4k_frame or placement or images or pool or golden" > "$OUT/pytest_encode.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_encode.log"; tail -6 "$OUT/pytest_encode.log"; rm -f gpucore.* core.*
echo "== flat content encode, 256 frames: summary passes (G2=0) / state look-back"
for K in uiflat constant sprite_alpha; do for G in 0 1; do
  KIND=$K QOIMI_ENC_G2=$G timeout 300 python tools/measure/enc_time.py - 256 2>&1 | tail -1 | sed "s/^/$K g2=$G /"; rm -f gpucore.* core.*
done; done | tee "$OUT/enc_g2.txt"
echo "== the same, 1024 frames"
for K in uiflat constant; do for G in 0 1; do
  KIND=$K QOIMI_ENC_G2=$G timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K g2=$G /"; rm -f gpucore.* core.*
done; done | tee -a "$OUT/enc_g2.txt"
KIND=photo timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/photo g2=1 /" | tee -a "$OUT/enc_g2.txt"
echo "== one frame"
for WH in "3840 2160" "1280 720"; do set -- $WH
  for G in 0 1; do W=$1 H=$2 QOIMI_ENC_G2=$G timeout 200 python tools/measure/single_trace.py 200 enc 2>&1 | tail -1 | sed "s/^/$1x$2 g2=$G /"; done
done | tee "$OUT/single_g2.txt"
echo "== done"
