#!/bin/bash
# round 5, session 11: the PIPE experiment - a wavefront that encodes set after set (QOIMI_ENC_PERSIST) asks for its next set's ticket and
# look-back window in front of its current set's placement (QOIMI_ENC_PIPE=1).  Bytes first, then the clock.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s11
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== bytes: selectable paths, a fuzz campaign with the form forced"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "selectable" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -3 "$OUT/pytest.log"
QOIMI_ENC_PIPE=1 QOIMI_ENC_PERSIST=5 timeout 600 python tests/fuzz_encode.py --iters 60 --seed 31 --batch8-half --max-pixels 2500000 --form 1 2>&1 | tail -2 | tee "$OUT/fuzz_pipe.txt"
rm -f gpucore.* core.*
echo "== 1024 photographs: one set per wavefront (default) / K sets per wavefront / the same with the next set asked for ahead"
# n_units = 675 quads x 1024 images = 691200 workgroups of four sets
KIND=photo timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/default /" | tee "$OUT/enc_pipe.txt"
for K in 2 3 4; do
  P=$((691200 / K))
  KIND=photo QOIMI_ENC_PERSIST=$P timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/K=$K persist=$P /" | tee -a "$OUT/enc_pipe.txt"
  KIND=photo QOIMI_ENC_PERSIST=$P QOIMI_ENC_PIPE=1 timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/K=$K persist=$P pipe /" | tee -a "$OUT/enc_pipe.txt"
done
KIND=photo timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/default again /" | tee -a "$OUT/enc_pipe.txt"
echo "== done"
