#!/bin/bash
# round 5, session 6: flat images as spans (no pixel ring in dec_segments_rec<OCH, true>; the expander writes unaligned heads and tails);
# the state look-back's set size and grid.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s6
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== pytest: decode tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "flat or decode or 4k_frame or batch or mixed or hostile or decoder_batch_fuzz or record or selectable" > "$OUT/pytest_decode.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_decode.log"; tail -4 "$OUT/pytest_decode.log"; rm -f gpucore.* core.*
echo "== decode kernels, 1024 frames"
for K in uiflat; do for B in 0 1024 2048; do
  if [ $B = 0 ]; then KIND=$K timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K auto /"
  else KIND=$K QOIMI_SEG_BYTES=$B timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K B=$B /"; fi
done; done | tee "$OUT/dec_span.txt"
KIND=constant timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/constant auto /" | tee -a "$OUT/dec_span.txt"
KIND=sprite_alpha timeout 300 python tools/measure/dec_time.py - 256 2>&1 | tail -1 | sed "s/^/sprite_alpha 256 /" | tee -a "$OUT/dec_span.txt"
echo "== state look-back: slabs per set, grid (1024 frames)"
for K in constant uiflat; do for R in 2 4 8; do
  KIND=$K QOIMI_ENC_GEN_SLABS=$R timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K gen_slabs=$R /"
done; done | tee "$OUT/enc_g2_tune.txt"
for K in constant uiflat; do for D in 8 128; do
  KIND=$K QOIMI_ENC_GEN_GRID_DIV=$D timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K grid_div=$D /"
done; done | tee -a "$OUT/enc_g2_tune.txt"
KIND=sprite_alpha QOIMI_ENC_GEN_SLABS=4 timeout 300 python tools/measure/enc_time.py - 256 2>&1 | tail -1 | sed "s/^/sprite_alpha 256 gen_slabs=4 /" | tee -a "$OUT/enc_g2_tune.txt"
echo "== done"
