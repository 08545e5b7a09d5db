#!/bin/bash
# round 6, session 36: campaigns on the final decode paths (class-by-class calls, passes that stop at a fixed point, small segments, the adaptive small-call path)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${SESSION:-r06_s36}
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 500 python tests/fuzz_decode_batch.py --iters 2500 --seed 6202 2>&1 | tail -1 | tee "$OUT/campaigns.txt"
timeout 300 python tests/fuzz_decode.py --iters 4000 --seed 6203 2>&1 | tail -1 | tee -a "$OUT/campaigns.txt"
timeout 400 python tests/fuzz_encode.py --iters 3000 --seconds 200 --seed 6201 --batch8-half 2>&1 | tail -1 | tee -a "$OUT/campaigns.txt"
timeout 600 python tests/stress_threads.py --threads 3 --calls 300 --batch-frames 1024 --default-placement 2>&1 | tail -1 | tee -a "$OUT/campaigns.txt"
python tests/fuzz/make_corpus.py /tmp/corpus > /dev/null 2>&1
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 700 tests/_bin/qoi_fuzz_diff -runs=15000 -rss_limit_mb=8192 -max_len=8192 -seed=20261001 -timeout=60 -print_final_stats=1 /tmp/corpus > "$OUT/fuzz_diff.log" 2>&1
echo "rc=$?" >> "$OUT/fuzz_diff.log"; grep -E "decoded by both|MISMATCH|ERROR|rc=|number_of_executed_units" "$OUT/fuzz_diff.log" | tail -5 | tee -a "$OUT/campaigns.txt"
echo "== done"
