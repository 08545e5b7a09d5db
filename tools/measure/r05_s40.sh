#!/bin/bash
# round 5, session 40: hostile streams through qoimi_decode_batch on the last commit (the round's last GPU seconds).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s40
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 50 python tests/fuzz_decode_batch.py --iters 300 --seed 4001 2>&1 | tail -1 | tee "$OUT/fuzz_dec.txt"
echo "== done"
