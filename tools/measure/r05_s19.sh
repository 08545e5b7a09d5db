#!/bin/bash
# round 5, session 19: long runs of the non-flat images as spans that the following QOI_OP_RUNs lengthen (dec_segments_rec); the one-pass hint
# against how far the device has come.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s19
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== pytest"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "small_calls or decode or gigabytes or hostile or mixed or batch or sweep or round_trip" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
timeout 300 python tests/fuzz_decode_batch.py --iters 150 --seed 91 2>&1 | tail -1 | tee "$OUT/fuzz_dec.txt"; rm -f gpucore.* core.*
echo "== single frames, the library's choice"
for K in photo constant uiflat sprite_alpha; do KIND=$K timeout 200 python tools/measure/single_trace.py 200 enc 2>&1 | tail -1 | sed "s/^/$K 4K auto /"; done | tee "$OUT/single_auto.txt"
echo "== decode per kernel, 256 frames"
for K in sprite_alpha photo photo_hard uiflat; do KIND=$K timeout 300 python tools/measure/dec_time.py - 256 2>&1 | tail -1 | sed "s/^/$K /"; done | tee "$OUT/dec_kernels.txt"
echo "== done"
