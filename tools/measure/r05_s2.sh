#!/bin/bash
# round 5, session 2: run descriptors (flat images: merged runs; other images: one per long QOI_OP_RUN) - parity first, then what they buy.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s2
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest: decode + new tests"
timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 900 -k "flat or decode or 4k_frame or gives_up or outgrows or batch or mixed or hostile or fuzz_short or record" > "$OUT/pytest_decode.log" 2>&1; echo "rc=$?" >> "$OUT/pytest_decode.log"; tail -4 "$OUT/pytest_decode.log"
echo "== decode kernels, 256 frames: descriptors off / flat only / all"
for K in uiflat constant sprite_alpha photo; do for D in 0 2; do
  KIND=$K QOIMI_DEC_RUN_DESC=$D timeout 300 python tools/measure/dec_time.py - 256 2>&1 | tail -1 | sed "s/^/$K desc=$D /"
done; done | tee "$OUT/dec_desc.txt"
echo "== flat content, 1024 frames: segment size"
for K in uiflat constant; do for B in 0 1024 4096; do
  if [ $B = 0 ]; then KIND=$K timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K auto /"
  else KIND=$K QOIMI_SEG_BYTES=$B timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K B=$B /"; fi
done; done | tee "$OUT/dec_flat_seg.txt"
echo "== full suite"
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
echo "== done"
