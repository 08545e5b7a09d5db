#!/bin/bash
# round 5, session 18: the one-pass hint as a window of 64 calls (pipelines of asynchronous calls); record arena of 24 GiB; where the decode
# of soft-alpha sprites spends its time.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s18
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== pytest"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "small_calls or granules or gigabytes or selectable" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"; rm -f gpucore.* core.*
echo "== single frames, the library's choice"
for K in photo constant uiflat sprite_alpha; do KIND=$K timeout 200 python tools/measure/single_trace.py 200 enc 2>&1 | tail -1 | sed "s/^/$K 4K auto /"; done | tee "$OUT/single_auto.txt"
echo "== decode per kernel, 256 frames"
for K in sprite_alpha noise photo; do KIND=$K timeout 300 python tools/measure/dec_time.py - 256 2>&1 | tail -1 | sed "s/^/$K /"; done | tee "$OUT/dec_kernels.txt"
KIND=sprite_alpha QOIMI_DEC_RUN_DESC=0 timeout 300 python tools/measure/dec_time.py - 256 2>&1 | tail -1 | sed "s/^/sprite_alpha run_desc=0 /" | tee -a "$OUT/dec_kernels.txt"
KIND=sprite_alpha QOIMI_DEC_RUN_DESC=1 timeout 300 python tools/measure/dec_time.py - 256 2>&1 | tail -1 | sed "s/^/sprite_alpha run_desc=1 /" | tee -a "$OUT/dec_kernels.txt"
echo "== done"
