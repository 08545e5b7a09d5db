#!/bin/bash
# round 5, session 17: small calls choose one pass or two by what the previous call met; set size of the pass over flagged images on content
# of more than a byte per pixel; the decoder's record arena capped (sub-batches) against device memory held.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s17
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== pytest"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "small_calls or granules or flat_frames or 4k_frame or gives_up" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
echo "== single frames, the library's choice (second and later calls)"
for K in photo constant uiflat sprite_alpha; do KIND=$K timeout 200 python tools/measure/single_trace.py 100 enc 2>&1 | tail -1 | sed "s/^/$K 4K auto /"; done | tee "$OUT/single_auto.txt"
echo "== slabs per set of the pass over flagged images (512 frames)"
for K in sprite_alpha uiflat; do for R in 8 4 3 2; do
  KIND=$K QOIMI_ENC_GEN_SLABS=$R timeout 300 python tools/measure/enc_time.py - 512 2>&1 | tail -1 | sed "s/^/$K 512 gen_slabs=$R /"
done; done | tee "$OUT/enc_gen_slabs.txt"
echo "== decode of 1024 photographs: record arena cap"
for CAP in 49152 24576 16384 12288 8192; do
  QOIMI_DEC_REC_CAP_MB=$CAP timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/cap_mb=$CAP /"
done | tee "$OUT/dec_cap.txt"
echo "== done"
