#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
for shape in "640 360" "1280 720" "1920 1080" "3840 2160" "7680 4320" "16384 16384"; do set -- $shape
echo -n "$1x$2 default: "; W=$1 H=$2 python tools/measure/single_trace.py 100 enc 2>&1 | grep "single frame"; done
for lb in 0 2; do echo -n "7680x4320 lookback=$lb R=3: "; W=7680 H=4320 QOIMI_ENC_LOOKBACK=$lb QOIMI_ENC_SET_SLABS=3 python tools/measure/single_trace.py 100 enc 2>&1 | grep "single frame"; done
