#!/bin/bash
# round 5, session 12: encode_step without its short-chunk tail where the long-chunk body has staged everything (-DQOIMI_ENC_SKIP_EMPTY_TAIL,
# build/exp_skiptail) - photographs must not pay, content dense in QOI_OP_RGB / QOI_OP_RGBA should gain.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s12
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
X="$PWD/build/exp_skiptail/libqoi_mi355x.so"
echo "== bytes (the experimental build): encode tests"
QOIMI_LIB=$X timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "4k_frame or sweep or fuzz_short or golden_byte" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -3 "$OUT/pytest.log"; rm -f gpucore.* core.*
for rep in 1 2; do
for K in photo photo_hard; do
  KIND=$K timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K 1024 default /"
  KIND=$K timeout 300 python tools/measure/enc_time.py $X 1024 2>&1 | tail -1 | sed "s/^/$K 1024 skiptail /"
done; done | tee "$OUT/enc_skiptail.txt"
for K in noise sprite_alpha; do
  KIND=$K timeout 300 python tools/measure/enc_time.py - 256 2>&1 | tail -1 | sed "s/^/$K 256 default /"
  KIND=$K timeout 300 python tools/measure/enc_time.py $X 256 2>&1 | tail -1 | sed "s/^/$K 256 skiptail /"
done | tee -a "$OUT/enc_skiptail.txt"
echo "== done"
