#!/bin/bash
# session 12, call 1: parity of the encoder's short-run path, encode A/B, P4 long-run stores A/B
cd "${GRAFT_REPO_ROOT:-/root/repo}"; O=gpurun_out/s12; mkdir -p $O; export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests -m gpu -q --timeout 600 -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  for lib in qoi_amd/lib/libqoi_mi355x.so build/exp_encold/libqoi_mi355x.so; do timeout 300 python tools/measure/enc_time.py $lib 256; done
done 2>&1 | tee $O/enc_ab.txt
for lib in qoi_amd/lib/libqoi_mi355x.so build/exp_encold/libqoi_mi355x.so; do timeout 300 python tools/measure/enc_time.py $lib 1024; done 2>&1 | tee -a $O/enc_ab.txt
for k in noise uiflat constant; do for lib in qoi_amd/lib/libqoi_mi355x.so build/exp_encold/libqoi_mi355x.so; do KIND=$k timeout 300 python tools/measure/enc_time.py $lib 256; done; done 2>&1 | tee -a $O/enc_ab.txt
for k in constant uiflat; do for i in 1 2; do for lib in qoi_amd/lib/libqoi_mi355x.so build/exp_splatbase/libqoi_mi355x.so; do KIND=$k timeout 300 python tools/measure/dec_time.py $lib 256; done; done; done 2>&1 | tee $O/dec_splat_ab.txt
