#!/bin/bash
# round 5, session 29: behind a batch of flagged images only, no first pass at all - every image through the pass over flagged images.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s29
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "flagged_images_only or granules or flat_frames or scratch_pool or fuzz" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
for A in 1 0; do
  for K in constant uiflat; do KIND=$K QOIMI_ENC_ALL_G2=$A timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K 1024 all_g2=$A /"; done
  KIND=sprite_alpha QOIMI_ENC_ALL_G2=$A timeout 300 python tools/measure/enc_time.py - 512 2>&1 | tail -1 | sed "s/^/sprite_alpha 512 all_g2=$A /"
done | tee "$OUT/enc_all_g2.txt"
echo "== done"
