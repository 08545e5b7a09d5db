#!/bin/bash
# round 6, session 52: with passes that stop at a fixed point, how many to launch: QOIMI_DEC_INNER1 (first round of flat images) x QOIMI_DEC_INNER (repair rounds)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s52
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for I1 in 3 6 10; do for I in 8 16; do
  for F in 32 256 1024; do
    QOIMI_DEC_INNER1=$I1 QOIMI_DEC_INNER=$I KIND=uiflat REPS=3 timeout 300 python tools/measure/dec_time.py - $F 2>&1 | tail -1 | sed -E "s/^.*'decode_total'/total/" | sed "s/^/inner1=$I1 inner=$I uiflat F=$F /"
  done
  QOIMI_DEC_INNER1=$I1 QOIMI_DEC_INNER=$I W=3840 H=2160 KIND=uiflat timeout 120 python tools/measure/single_trace.py 40 dec 2>&1 | tail -1 | sed "s/^/inner1=$I1 inner=$I lone 4K uiflat /"
  QOIMI_DEC_INNER1=$I1 QOIMI_DEC_INNER=$I timeout 300 python tools/measure/mixed_trace.py 2>&1 | tail -2 | head -1 | cut -c60-100 | sed "s/^/inner1=$I1 inner=$I mixed /"
done; done | tee "$OUT/inner.txt"
