#!/bin/bash
# round 5, session 1: the whole GPU suite (new content classes, hash check, 60 s batch fuzz, per-workgroup tree tickets), the bench line with
# every class split into encode / decode, the encoder's set size on the new classes, single-frame encode with and without the tree's tickets.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s1
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest -m gpu"; timeout 1500 python -m pytest tests -m gpu -q -x --timeout 900 > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
echo "== bench"; timeout 900 python bench.py > "$OUT/bench.log" 2>&1; echo "rc=$?" >> "$OUT/bench.log"; tail -c 600 "$OUT/bench.log"
echo "== set size on the new classes (128 frames)"
for K in photo_hard sprite_alpha; do for R in 1 2 3; do
  KIND=$K QOIMI_ENC_SET_SLABS=$R timeout 200 python tools/measure/enc_time.py - 128 2>&1 | tail -1 | sed "s/^/$K R=$R /"
done; done | tee "$OUT/enc_setsize.txt"
echo "== decode kernels on the new classes (128 frames)"
for K in photo photo_hard sprite_alpha; do KIND=$K timeout 200 python tools/measure/dec_time.py - 128 2>&1 | tail -1 | sed "s/^/$K /"; done | tee "$OUT/dec_classes.txt"
echo "== one frame: tree placement with per-workgroup tickets (default) and by workgroup index"
for WH in "3840 2160" "1280 720"; do set -- $WH
  for T in 1 0; do W=$1 H=$2 QOIMI_ENC_TICKET=$T timeout 200 python tools/measure/single_trace.py 200 enc 2>&1 | tail -1 | sed "s/^/$1x$2 ticket=$T /"; done
done | tee "$OUT/single_ticket.txt"
echo "== done"
