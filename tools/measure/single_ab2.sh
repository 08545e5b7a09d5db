#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
for shape in "640 360" "1280 720" "1920 1080" "2560 1440" "5120 2880"; do set -- $shape
for lb in 0 1 2; do for r in 1 2 3; do echo -n "$1x$2 lookback=$lb R=$r: "; W=$1 H=$2 QOIMI_ENC_LOOKBACK=$lb QOIMI_ENC_SET_SLABS=$r python tools/measure/single_trace.py 200 enc 2>&1 | grep "single frame"; done; done; done
