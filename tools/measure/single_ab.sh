#!/bin/bash
# single-image encode / decode wall clock per placement form and set size: tools/measure/single_ab.sh
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
for lb in 0 2 1; do for r in 1 2 3; do echo -n "4K lookback=$lb R=$r: "; QOIMI_ENC_LOOKBACK=$lb QOIMI_ENC_SET_SLABS=$r python tools/measure/single_trace.py 200 enc 2>&1 | grep "single frame"; done; done
for lb in 0 2; do for r in 2 3 4; do echo -n "16K lookback=$lb R=$r: "; W=16384 H=16384 QOIMI_ENC_LOOKBACK=$lb QOIMI_ENC_SET_SLABS=$r python tools/measure/single_trace.py 30 enc 2>&1 | grep "single frame"; done; done
