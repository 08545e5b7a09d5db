#!/bin/bash
# round 5, session 15: ONE flat frame - the state look-back with every set of the image in flight at once (tree placement) against the summary passes.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s15
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
for K in constant uiflat sprite_alpha; do for WH in "3840 2160" "1280 720" "7680 4320"; do set -- $WH
  for G in 0 1; do KIND=$K W=$1 H=$2 QOIMI_ENC_G2=$G timeout 200 python tools/measure/single_trace.py 100 enc 2>&1 | tail -1 | sed "s/^/$K $1x$2 g2=$G /"; done
done; done | tee "$OUT/single_flat.txt"
echo "== done"
