#!/bin/bash
# round 5, session 22: sets of up to 16 slabs in the pass over flagged images.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s22
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
for R in 8 12 16; do
  for K in constant uiflat; do KIND=$K QOIMI_ENC_GEN_SLABS=$R timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K 1024 gen_slabs=$R /"; done
  KIND=sprite_alpha QOIMI_ENC_GEN_SLABS=$R timeout 300 python tools/measure/enc_time.py - 512 2>&1 | tail -1 | sed "s/^/sprite_alpha 512 gen_slabs=$R /"
done | tee "$OUT/enc_gen_slabs16.txt"
QOIMI_ENC_GEN_SLABS=16 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "flat or granules or images or letterbox or start or small_calls" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
QOIMI_ENC_GEN_SLABS=16 timeout 400 python tests/fuzz_encode.py --seconds 20 --seed 80 --batch8-half 2>&1 | tail -1 | tee "$OUT/fuzz.txt"; rm -f gpucore.* core.*
echo "== done"
