#!/bin/bash
# round 5, session 20: the state look-back walks a set's last two groups first and the groups in front of them only where those do not
# write all 64 slots (flat stretches).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s20
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== pytest: encode tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "flat or fuzz or selectable or sweep or 4k_frame or granules or images or letterbox or start or small_calls" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
timeout 400 python tests/fuzz_encode.py --seconds 30 --seed 78 --batch8-half 2>&1 | tail -1 | tee "$OUT/fuzz.txt"; rm -f gpucore.* core.*
echo "== batches (1024 frames; sprite / photo_hard 512)"
for K in photo constant uiflat; do KIND=$K timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K 1024 /"; done | tee "$OUT/enc_tail_first.txt"
KIND=sprite_alpha timeout 300 python tools/measure/enc_time.py - 512 2>&1 | tail -1 | sed "s/^/sprite_alpha 512 /" | tee -a "$OUT/enc_tail_first.txt"
KIND=sprite_alpha QOIMI_ENC_UNI=1 timeout 300 python tools/measure/enc_time.py - 512 2>&1 | tail -1 | sed "s/^/sprite_alpha 512 uni=1 /" | tee -a "$OUT/enc_tail_first.txt"
echo "== single frames"
for K in constant uiflat sprite_alpha; do for U in 0 1; do KIND=$K QOIMI_ENC_UNI=$U timeout 200 python tools/measure/single_trace.py 200 enc 2>&1 | tail -1 | sed "s/^/$K 4K uni=$U /"; done; done | tee "$OUT/single.txt"
echo "== done"
