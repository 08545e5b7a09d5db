#!/bin/bash
# round 4, session 2: SPREAD as default + the cheaper spin path (parity first), then what a persistent grid costs with and without the
# copy-out stores (is it the in-order store acknowledgement in front of the next unit's ticket?), phase stamps with re-poll counts
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-s2}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
SEL="encode or sweep or selectable or mixed or flat_frames or set_sizes or one_context or three_channel or 4k_frame or batch_1080p or many_small"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SEL" > $OUT/pytest_enc.log 2>&1; echo "rc=$?" >> $OUT/pytest_enc.log; tail -3 $OUT/pytest_enc.log
for e in "" "QOIMI_ENC_SPREAD=0" "QOIMI_ENC_PERSIST=1536"; do
  for f in 1024; do env $e python tools/dev/enc_time.py - $f 2>&1 | grep -v amdgpu.ids | sed "s/^/[$e] $f frames: /"; done
done | tee $OUT/enc_time.txt
for e in "" "QOIMI_ENC_PERSIST=1536" "QOIMI_ENC_PERSIST=1024"; do
  env $e python tools/dev/enc_time.py build/exp_nocopy/libqoi_mi355x.so 1024 2>&1 | grep -v amdgpu.ids | sed "s/^/[nocopy $e] 1024 frames: /"
done | tee -a $OUT/enc_time.txt
(python tools/dev/enc_phases.py build/exp_phases/libqoi_mi355x.so 256; QOIMI_ENC_SPREAD=0 python tools/dev/enc_phases.py build/exp_phases/libqoi_mi355x.so 256) 2>&1 | grep -v amdgpu.ids | tee $OUT/phases.txt
