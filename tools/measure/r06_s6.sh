#!/bin/bash
# round 6, session 6: one 4K frame's encode by set size / placement / tickets / one pass (wall clock of 400 calls each)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s6
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for T in 1 0; do for U in 0 1; do for S in 0 1 2 3 4; do
  QOIMI_ENC_TREE_TICKET=$T QOIMI_ENC_UNI=$U QOIMI_ENC_SET_SLABS=$S timeout 100 python tools/measure/single_trace.py 400 enc 2>&1 | tail -1 | sed "s/^/ticket=$T uni=$U slabs=$S /"
done; done; done | tee "$OUT/single_enc_matrix.txt"
for L in 1 0; do QOIMI_ENC_LOOKBACK=$L timeout 100 python tools/measure/single_trace.py 400 enc 2>&1 | tail -1 | sed "s/^/lookback=$L /"; done | tee -a "$OUT/single_enc_matrix.txt"
