#!/bin/bash
# First GPU call of the next round ("Next round"): the two knobs prepared at the end of round 3, parity first.
#   QOIMI_ENC_SPREAD=1   wavefronts of a workgroup on consecutive images (never run on a GPU before)
#   QOIMI_ENC_PERSIST=N  the default kernel with N persistent workgroups (does persistence alone cost the 30 % of the deferred forms?)
# and the phase stamps of a set under SPREAD (build the diagnostic library first, on the CPU:
#   mkdir -p build/exp_phases && cd qoi_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function \
#     -Wno-int-to-pointer-cast -DQOIMI_ENC_PHASES -c qoi_encode.hip -o ../../build/exp_phases/enc.o && hipcc --offload-arch=gfx950 -shared -fPIC \
#     -o ../../build/exp_phases/libqoi_mi355x.so ../../build/exp_phases/enc.o ../lib/obj/qoi_host.o ../lib/obj/qoi_decode.o ../lib/obj/qoi_synth.o )
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-next}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
SEL="encode or sweep or selectable or mixed or flat_frames or set_sizes or one_context or three_channel or 4k_frame or batch_1080p or many_small"
QOIMI_ENC_SPREAD=1 timeout 250 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SEL" > $OUT/pytest_spread.log 2>&1; echo "rc=$?" >> $OUT/pytest_spread.log; tail -3 $OUT/pytest_spread.log
for e in "" "QOIMI_ENC_SPREAD=1" "QOIMI_ENC_PERSIST=1536" "QOIMI_ENC_PERSIST=1280" "QOIMI_ENC_PERSIST=1024" "QOIMI_ENC_SPREAD=1 QOIMI_ENC_PERSIST=1536" "QOIMI_ENC_TICKET=0" "QOIMI_ENC_SPREAD=1 QOIMI_ENC_SET_SLABS=2" "QOIMI_ENC_SPREAD=1 QOIMI_ENC_SET_SLABS=4" "QOIMI_ENC_SET_SLABS=2"; do
  for f in 512 1024; do env $e python tools/dev/enc_time.py - $f 2>&1 | grep -v amdgpu.ids | sed "s/^/[$e] $f frames: /"; done
done | tee $OUT/enc_time.txt
if [ -f build/exp_phases/libqoi_mi355x.so ]; then
  (python tools/dev/enc_phases.py build/exp_phases/libqoi_mi355x.so 256; QOIMI_ENC_SPREAD=1 python tools/dev/enc_phases.py build/exp_phases/libqoi_mi355x.so 256) 2>&1 | grep -v amdgpu.ids | tee $OUT/phases.txt
fi
