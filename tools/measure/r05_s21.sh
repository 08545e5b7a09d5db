#!/bin/bash
# round 5, session 21: tail-first walk as ONE pipeline (the first group in front asked for before the tail is looked at); slabs per set of the
# pass over flagged images again, now that dense sets walk two groups only.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s21
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== pytest: encode tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "flat or fuzz or selectable or granules or images or letterbox or start or small_calls" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
timeout 400 python tests/fuzz_encode.py --seconds 20 --seed 79 --batch8-half 2>&1 | tail -1 | tee "$OUT/fuzz.txt"; rm -f gpucore.* core.*
echo "== batches (1024 frames; sprite 512)"
for K in constant uiflat; do KIND=$K timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K 1024 /"; done | tee "$OUT/enc_tail_first.txt"
for R in 8 6 4 3; do KIND=sprite_alpha QOIMI_ENC_GEN_SLABS=$R timeout 300 python tools/measure/enc_time.py - 512 2>&1 | tail -1 | sed "s/^/sprite_alpha 512 gen_slabs=$R /"; done | tee -a "$OUT/enc_tail_first.txt"
echo "== done"
