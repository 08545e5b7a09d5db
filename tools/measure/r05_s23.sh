#!/bin/bash
# round 5, session 23: first pass with a sixteenth of its workgroups behind a batch of flagged images only; sixteen slabs per set of the pass over
# flagged images as the default of large calls.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s23
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "flagged_images_only or granules or flat_frames or selectable" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
for K in photo constant uiflat; do KIND=$K timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K 1024 /"; done | tee "$OUT/enc_first_pass.txt"
for K in sprite_alpha photo_hard; do KIND=$K timeout 300 python tools/measure/enc_time.py - 512 2>&1 | tail -1 | sed "s/^/$K 512 /"; done | tee -a "$OUT/enc_first_pass.txt"
echo "== done"
