#!/bin/bash
# round 4, session 3: instruction-rate table (which VALU ops run at the fast rate), copy-out with unaligned 16-byte stores (parity
# first), wave priority experiments
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-s3}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
./build/ubench/valu_tput2 > $OUT/valu_tput2.txt 2>&1; tail -40 $OUT/valu_tput2.txt
SEL="encode or sweep or selectable or mixed or flat_frames or set_sizes or one_context or three_channel or 4k_frame or batch_1080p or many_small or recheck"
QOIMI_LIB=build/exp_ua/libqoi_mi355x.so timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SEL" > $OUT/pytest_ua.log 2>&1; echo "rc=$?" >> $OUT/pytest_ua.log; tail -3 $OUT/pytest_ua.log
timeout 100 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "recheck" > $OUT/pytest_recheck.log 2>&1; echo "rc=$?" >> $OUT/pytest_recheck.log; tail -3 $OUT/pytest_recheck.log
for arm in "-:" "build/exp_ua/libqoi_mi355x.so:" "build/exp_prio1/libqoi_mi355x.so:" "build/exp_prio2/libqoi_mi355x.so:QOIMI_ENC_PERSIST=1536" "build/exp_prio2/libqoi_mi355x.so:" "-:"; do
  lib=${arm%%:*}; e=${arm#*:}
  env $e python tools/dev/enc_time.py $lib 1024 2>&1 | grep -v amdgpu.ids | sed "s|^|[$lib $e] 1024 frames: |"
done | tee $OUT/enc_time.txt
