#!/bin/bash
# round 5, session 25: campaigns on the final build - encoder fuzz (half of the calls with eight images and more), hostile streams through
# qoimi_decode_batch, the libFuzzer + ASan differential decode harness; the pool test with its new bound.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s25
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "scratch_pool or flagged_images_only or small_calls" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -3 "$OUT/pytest.log"
timeout 400 python tests/fuzz_encode.py --seconds 90 --seed 2025 --batch8-half 2>&1 | tail -1 | tee "$OUT/fuzz_encode.txt"; rm -f gpucore.* core.*
timeout 400 python tests/fuzz_decode_batch.py --iters 600 --seed 2026 2>&1 | tail -1 | tee "$OUT/fuzz_decode_batch.txt"; rm -f gpucore.* core.*
python tests/fuzz/make_corpus.py /tmp/corpus > /dev/null 2>&1
ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1" UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1" timeout 200 tests/_bin/qoi_fuzz_diff -runs=30000 -rss_limit_mb=8192 -max_len=8192 -seed=13 -timeout=60 -max_total_time=150 -print_final_stats=1 /tmp/corpus > "$OUT/fuzz_diff_full.txt" 2>&1; echo "rc=$?" >> "$OUT/fuzz_diff_full.txt"
(head -4 "$OUT/fuzz_diff_full.txt"; grep -c "MISMATCH\|ERROR: AddressSanitizer\|runtime error" "$OUT/fuzz_diff_full.txt"; tail -12 "$OUT/fuzz_diff_full.txt") > "$OUT/fuzz_diff.txt"; rm -f "$OUT/fuzz_diff_full.txt"; tail -8 "$OUT/fuzz_diff.txt"
echo "== done"
