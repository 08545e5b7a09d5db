#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-s5}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for arm in "-:" "-:QOIMI_EXP_POOL_SMALL=1" "build/exp_noatom/libqoi_mi355x.so:" "build/exp_plain/libqoi_mi355x.so:" "build/exp_both/libqoi_mi355x.so:QOIMI_EXP_POOL_SMALL=1" "-:QOIMI_ENC_LOOKBACK=0"; do
  lib=${arm%%:*}; e=${arm#*:}
  env $e KIND=noise python tools/dev/enc_time.py $lib 256 2>&1 | grep -v amdgpu.ids | sed "s|^|[noise $lib $e] 256 frames: |"
done | tee $OUT/enc_time.txt
