#!/bin/bash
# enc_time.py over a list of builds, twice, 256 frames: tools/measure/ab_libs.sh lib...
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
for i in 1 2; do for lib in "$@"; do timeout 300 python tools/measure/enc_time.py $lib ${FRAMES:-256} 2>&1 | grep -v amdgpu.ids; done; done
