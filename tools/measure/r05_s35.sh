#!/bin/bash
# round 5, session 35: campaigns on the round's last build - encoder fuzz (batches of eight and more in half of the calls; odd strides; the
# drop-in entry points), hostile streams through qoimi_decode_batch, twelve threads of single-frame encodes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s35
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
{
timeout 300 python tests/fuzz_encode.py --iters 800 --seconds 75 --seed 3101 --batch8-half 2>&1 | tail -1
timeout 200 python tests/fuzz_encode.py --iters 300 --seconds 30 --seed 3102 --odd-strides --max-pixels 2500000 2>&1 | tail -1
timeout 200 python tests/fuzz_encode.py --iters 300 --seconds 30 --seed 3103 --dropin 2>&1 | tail -1
timeout 300 python tests/fuzz_decode_batch.py --iters 700 --seed 3104 2>&1 | tail -1
timeout 300 python tests/stress_threads.py --threads 12 --calls 400 2>&1 | tail -2
} | tee "$OUT/campaigns.txt"
rm -f gpucore.* core.*
echo "== done"
