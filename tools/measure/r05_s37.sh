#!/bin/bash
# round 5, session 37: calls of a few images without their hipMemsetAsync - two regions of records used in turn, the first launch of a call zeroes
# the next call's (QOIMI_ENC_PREZERO=0: the memset as before).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s37
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
for K in photo constant; do for Z in 1 0; do KIND=$K QOIMI_ENC_PREZERO=$Z timeout 100 python tools/measure/single_trace.py 300 enc 2>&1 | tail -1 | sed "s/^/$K 4K prezero=$Z /"; done; done | tee "$OUT/single_prezero.txt"
KIND=photo W=1280 H=720 QOIMI_ENC_PREZERO=1 timeout 100 python tools/measure/single_trace.py 300 enc 2>&1 | tail -1 | sed "s/^/photo 720p prezero=1 /" | tee -a "$OUT/single_prezero.txt"
KIND=photo W=1280 H=720 QOIMI_ENC_PREZERO=0 timeout 100 python tools/measure/single_trace.py 300 enc 2>&1 | tail -1 | sed "s/^/photo 720p prezero=0 /" | tee -a "$OUT/single_prezero.txt"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_threads.py tests/test_c_dropin.py -m gpu -q -x --timeout 600 -k "4k_frame or flat_frames or small_calls or selectable or granules or gives_up or images or threads or dropin or letterbox or start or sweep" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -3 "$OUT/pytest.log"; rm -f gpucore.* core.*
timeout 200 python tests/fuzz_encode.py --iters 400 --seconds 25 --seed 3701 --batch8-half 2>&1 | tail -1 | tee "$OUT/fuzz.txt"
timeout 200 python tests/stress_threads.py --threads 8 --calls 200 2>&1 | tail -1 | tee -a "$OUT/fuzz.txt"
echo "== done"
