#!/bin/bash
# round 6, session 21: libFuzzer + ASan differential campaign on the round's host shim and kernels; large single images on the small-call path
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s21
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
python tests/fuzz/make_corpus.py /tmp/corpus > /dev/null 2>&1
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 900 tests/_bin/qoi_fuzz_diff -runs=20000 -rss_limit_mb=8192 -max_len=8192 -seed=20260930 -timeout=60 -print_final_stats=1 /tmp/corpus > "$OUT/fuzz_diff.log" 2>&1
echo "rc=$?" >> "$OUT/fuzz_diff.log"; grep -E "decoded by both|MISMATCH|ERROR|rc=|number_of_executed_units" "$OUT/fuzz_diff.log" | tail -5 | tee "$OUT/fuzz_diff.txt"
for S in "8192 8192" "16384 16384" "5120 2880" "2560 1440"; do set -- $S; W=$1 H=$2 timeout 120 python tools/measure/single_trace.py 30 both 2>&1 | tail -1 | sed "s/^/$1x$2 /"; done | tee "$OUT/single_sizes.txt"
