#!/bin/bash
# round 6, session 18: campaigns on the round's kernels - single frames beside 1024-frame batches, the encoder fuzz, hostile streams through qoimi_decode_batch
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s18
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 900 python tests/stress_threads.py --threads 3 --calls 400 --batch-frames 1024 --default-placement 2>&1 | tail -1 | tee "$OUT/stress_batches.txt"
timeout 600 python tests/stress_threads.py --threads 8 --calls 200 2>&1 | tail -1 | tee -a "$OUT/stress_batches.txt"
timeout 400 python tests/fuzz_encode.py --iters 3000 --seconds 240 --seed 6001 --batch8-half 2>&1 | tail -1 | tee "$OUT/campaigns.txt"
timeout 300 python tests/fuzz_decode_batch.py --iters 1500 --seed 6002 2>&1 | tail -1 | tee -a "$OUT/campaigns.txt"
timeout 300 python tests/fuzz_decode.py --iters 4000 --seed 6003 2>&1 | tail -1 | tee -a "$OUT/campaigns.txt"
echo "== done"
