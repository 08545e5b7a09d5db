#!/bin/bash
# round 5, session 33: encode calls of bench.py's class loop one by one (constant read 10.7-14.6 ms per call in session 32, 6.4-7.1 everywhere else).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s33
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
ASYNC=1 timeout 600 python tools/measure/enc_seq.py 1024 noise constant uiflat constant sprite_alpha 2>&1 | tee "$OUT/enc_seq_async.txt" | cut -c1-220
echo "== done"
