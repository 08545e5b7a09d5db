#!/bin/bash
# round 6, session 26: refinement passes that stop at a fixed point (QOIMI_DEC_CONV): mixed directory, lone flat frames, batches of UI frames
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s26
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for C in 0 1; do
  QOIMI_DEC_CONV=$C timeout 300 python tools/measure/mixed_trace.py 2>&1 | tail -2 | sed "s/^/conv=$C mixed /"
  for K in uiflat constant sprite_alpha; do for S in "3840 2160" "1280 720"; do set -- $S
    QOIMI_DEC_CONV=$C W=$1 H=$2 KIND=$K STATS=1 timeout 120 python tools/measure/single_trace.py 40 dec 2>&1 | tail -3 | sed "s/^/conv=$C $1x$2 $K: /"
  done; done
  for K in uiflat sprite_alpha photo; do
    QOIMI_DEC_CONV=$C timeout 300 python bench.py --frames 256 --steps 3 --warmup 1 --kind $K --no-cpu --no-others --no-single --no-configs 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('conv=$C batch256 $K', d['value'], d['ms_per_step'], d.get('decode_rounds'), d.get('kernel_ms_per_step'))"
  done
done 2>&1 | tee "$OUT/conv_ab.txt"
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee "$OUT/pytest_gpu.txt"
