#!/bin/bash
# round 6, session 27: segments below 128 bytes as the default of small calls: the GPU suite, the campaigns on hostile streams, lone frames by size and class
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s27
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/pytest_gpu.txt"
timeout 300 python tests/fuzz_decode_batch.py --iters 1500 --seed 6102 2>&1 | tail -1 | tee "$OUT/campaigns.txt"
timeout 300 python tests/fuzz_decode.py --iters 4000 --seed 6103 2>&1 | tail -1 | tee -a "$OUT/campaigns.txt"
python tests/fuzz/make_corpus.py /tmp/corpus > /dev/null 2>&1
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allocator_may_return_null=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 timeout 600 tests/_bin/qoi_fuzz_diff -runs=10000 -rss_limit_mb=8192 -max_len=8192 -seed=20260931 -timeout=60 -print_final_stats=1 /tmp/corpus > "$OUT/fuzz_diff.log" 2>&1
echo "rc=$?" >> "$OUT/fuzz_diff.log"; grep -E "decoded by both|MISMATCH|ERROR|rc=|number_of_executed_units" "$OUT/fuzz_diff.log" | tail -5 | tee "$OUT/fuzz_diff.txt"
export QOIMI_TUNING=1
for S in "3840 2160" "2560 1440" "1920 1080" "1280 720" "640 360"; do set -- $S; for K in photo uiflat constant; do
  W=$1 H=$2 KIND=$K timeout 120 python tools/measure/single_trace.py 40 dec 2>&1 | tail -1 | sed "s/^/$1x$2 $K /"
done; done | tee "$OUT/single_sizes.txt"
