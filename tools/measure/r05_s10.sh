#!/bin/bash
# round 5, session 10: the state look-back's own walk with one load per pixel; refinement rounds with eight repetitions; first-round
# refinement passes of flat images now that a round is cheap.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s10
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== pytest: encode tests (flat content, fuzz, granules across calls), qoibench tool"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_qoibench_cli.py -m gpu -q -x --timeout 900 -k "flat or fuzz or selectable or sweep or 4k_frame or granules or images or gpu_rows or letterbox or start" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
echo "== flat content encode (1024 frames; sprite 256)"
for K in constant uiflat; do KIND=$K timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K 1024 /"; done | tee "$OUT/enc_flat.txt"
KIND=sprite_alpha timeout 300 python tools/measure/enc_time.py - 256 2>&1 | tail -1 | sed "s/^/sprite_alpha 256 /" | tee -a "$OUT/enc_flat.txt"
echo "== uiflat decode, 1024 frames: first-round refinement passes"
for I in 3 2 1; do KIND=uiflat QOIMI_DEC_INNER1=$I timeout 300 python tools/measure/dec_time.py - 1024 2>&1 | tail -1 | sed "s/^/uiflat inner1=$I /"; done | tee "$OUT/dec_uiflat_inner1.txt"
echo "== done"
