#!/bin/bash
# round 6, session 29: a mixed call decoded class by class: the new test, the mixed directory by knob and segment size
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s29
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "mixed_call or many_small or gigabytes or shapes" 2>&1 | tail -4 | tee "$OUT/pytest_subset.txt"
for V in "QOIMI_DEC_CLASS_SPLIT=0" "QOIMI_DEC_CLASS_SPLIT=1" "QOIMI_SEG_BYTES=256" "QOIMI_SEG_BYTES=512" "QOIMI_SEG_BYTES=1024" "QOIMI_SEG_BYTES=2048"; do
  env $V timeout 300 python tools/measure/mixed_trace.py 2>&1 | tail -2 | sed "s/^/$V /"
done | cut -c1-1300 | tee "$OUT/mixed_split.txt"
