#!/bin/bash
# round 5, session 14: the transcoder's descriptor (whole read extent inside, base aligned down) on odd addresses; the grid of the pass over
# flagged images when the previous batch held some.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s14
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
echo "== decode tests, odd addresses and strides"
timeout 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "decode or gigabytes or hostile or mixed or batch" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -3 "$OUT/pytest.log"
timeout 600 python tests/fuzz_encode.py --iters 40 --seed 41 --odd-strides --max-pixels 2500000 2>&1 | tail -1 | tee "$OUT/fuzz_odd.txt"; rm -f gpucore.* core.*
echo "== grid of the pass over flagged images (1024 frames; sprite 512)"
for K in constant uiflat; do for D in 32 8 4 2 1; do
  KIND=$K QOIMI_ENC_GEN_GRID_HOT=$D timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/$K hot_div=$D /"
done; done | tee "$OUT/enc_grid.txt"
for D in 32 4 1; do KIND=sprite_alpha QOIMI_ENC_GEN_GRID_HOT=$D timeout 300 python tools/measure/enc_time.py - 512 2>&1 | tail -1 | sed "s/^/sprite_alpha 512 hot_div=$D /"; done | tee -a "$OUT/enc_grid.txt"
KIND=photo timeout 300 python tools/measure/enc_time.py - 1024 2>&1 | tail -1 | sed "s/^/photo /" | tee -a "$OUT/enc_grid.txt"
echo "== done"
