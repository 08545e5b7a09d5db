#!/bin/bash
# round 5, session 3: the run-descriptor expander as a queue of segments (16-byte descriptors) - parity, then flat content again.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s3
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest: decode tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "flat or decode or 4k_frame or batch or mixed or hostile or decoder_batch_fuzz or record" > "$OUT/pytest_decode.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_decode.log"; tail -4 "$OUT/pytest_decode.log"
echo "== decode kernels, 256 frames"
for K in uiflat constant sprite_alpha; do
  KIND=$K timeout 300 python tools/measure/dec_time.py - 256 2>&1 | tail -1 | sed "s/^/$K /"
done | tee "$OUT/dec_desc.txt"
echo "== flat content, 1024 frames: segment size"
for K in uiflat; do for B in 0 512 1024 2048; do
  if [ $B = 0 ]; then KIND=$K timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K auto /"
  else KIND=$K QOIMI_SEG_BYTES=$B timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K B=$B /"; fi
done; done | tee "$OUT/dec_flat_seg.txt"
KIND=constant timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/constant auto /" | tee -a "$OUT/dec_flat_seg.txt"
KIND=uiflat QOIMI_DEC_INNER1=1 timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/uiflat auto inner1=1 /" | tee -a "$OUT/dec_flat_seg.txt"
KIND=uiflat QOIMI_DEC_INNER1=0 timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/uiflat auto inner1=0 /" | tee -a "$OUT/dec_flat_seg.txt"
echo "== done"
