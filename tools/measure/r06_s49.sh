#!/bin/bash
# round 6, session 49: the GPU suite on the round's last build, then long campaigns: libFuzzer + ASan differential (qoi_decode against the unmodified reference), hostile streams in batches
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s49
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -3 | tee "$OUT/pytest.txt"
python tests/fuzz/make_corpus.py /tmp/corpus > /dev/null 2>&1
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 1500 tests/_bin/qoi_fuzz_diff -runs=40000 -rss_limit_mb=8192 -max_len=16384 -seed=20261002 -timeout=60 -print_final_stats=1 /tmp/corpus > "$OUT/fuzz_diff.log" 2>&1
echo "rc=$?" >> "$OUT/fuzz_diff.log"; grep -E "decoded by both|MISMATCH|ERROR|rc=|number_of_executed_units" "$OUT/fuzz_diff.log" | tail -5 | tee "$OUT/campaigns.txt"
timeout 900 python tests/fuzz_decode_batch.py --iters 6000 --seed 6502 2>&1 | tail -1 | tee -a "$OUT/campaigns.txt"
timeout 500 python tests/fuzz_encode.py --iters 5000 --seconds 400 --seed 6501 --batch8-half 2>&1 | tail -1 | tee -a "$OUT/campaigns.txt"
echo "== done"
