#!/bin/bash
# round 5, session 16: the state look-back publishing inclusive at once where a set knows its own entry (window of eight sets per poll);
# the ONE-pass encoder (QOIMI_ENC_UNI=1: a set whose warm window does not do takes the state look-back by itself) - parity, batches, single frames.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s16
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== pytest: encode tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "flat or fuzz or selectable or sweep or 4k_frame or granules or images or letterbox or start" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
QOIMI_ENC_UNI=1 timeout 400 python tests/fuzz_encode.py --seconds 40 --seed 77 --batch8-half 2>&1 | tail -1 | tee "$OUT/fuzz_uni.txt"; rm -f gpucore.* core.*
echo "== batches, one pass against two (1024 frames; sprite / photo_hard 512)"
for U in 0 1; do
  for K in photo constant uiflat; do KIND=$K QOIMI_ENC_UNI=$U timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K 1024 uni=$U /"; done
  for K in sprite_alpha photo_hard; do KIND=$K QOIMI_ENC_UNI=$U timeout 300 python tools/measure/enc_time.py - 512 2>&1 | tail -1 | sed "s/^/$K 512 uni=$U /"; done
done | tee "$OUT/enc_uni.txt"
echo "== single frames"
for K in photo constant uiflat sprite_alpha; do for WH in "3840 2160" "1280 720"; do set -- $WH
  for U in 0 1; do KIND=$K W=$1 H=$2 QOIMI_ENC_UNI=$U timeout 200 python tools/measure/single_trace.py 100 enc 2>&1 | tail -1 | sed "s/^/$K $1x$2 uni=$U /"; done
done; done | tee "$OUT/single_uni.txt"
echo "== done"
