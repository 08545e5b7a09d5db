#!/bin/bash
# round 5, session 39 (the round's last GPU seconds): encoder fuzz on the last commit, device buffers at odd addresses with odd strides.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s39
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 70 python tests/fuzz_encode.py --iters 400 --seconds 35 --seed 3901 --odd-strides --max-pixels 2500000 2>&1 | tail -1 | tee "$OUT/fuzz_odd.txt"
echo "== done"
