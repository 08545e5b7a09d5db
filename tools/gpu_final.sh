#!/bin/bash
# Last GPU call of a session: parity + timing of an encoder form that is a candidate for the default (CAND, QOIMI_ENC_CLS value), then
# tools/gpu_session.sh with the candidate exported if it passed and is at least 4 % faster on 512 frames - the session then measures
# what the tree is about to make its default.  Decision and numbers in gpurun_out/$1/decision.txt.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; NAME=${1:-final}; OUT=gpurun_out/$NAME; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
CAND=${CAND:-2}
SEL="encode or sweep or selectable or mixed or flat_frames or set_sizes or one_context or three_channel or 4k_frame or batch_1080p or many_small"
QOIMI_ENC_CLS=$CAND timeout 250 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$SEL" > $OUT/pytest_cand.log 2>&1; RC=$?; tail -3 $OUT/pytest_cand.log
(python tools/dev/enc_time.py - 512; QOIMI_ENC_CLS=$CAND python tools/dev/enc_time.py - 512; QOIMI_ENC_CLS=$CAND python tools/dev/enc_time.py - 1024; KIND=noise python tools/dev/enc_time.py - 128; KIND=noise QOIMI_ENC_CLS=$CAND python tools/dev/enc_time.py - 128) 2>&1 | grep -v amdgpu.ids | tee $OUT/enc_time.txt
USE=$(python - $OUT/enc_time.txt $RC <<'PY'
import re, sys
t = [float(m.group(1)) for m in re.finditer(r"'enc_slabs': ([0-9.]+)", open(sys.argv[1]).read())]
ok = int(sys.argv[2]) == 0 and len(t) >= 2 and t[1] < 0.96 * t[0]
print(1 if ok else 0)
PY
)
echo "candidate QOIMI_ENC_CLS=$CAND: tests rc=$RC, adopted=$USE" | tee $OUT/decision.txt
[ "$USE" = 1 ] && export QOIMI_ENC_CLS=$CAND
SESSION=$NAME SKIP_PMC=1 bash tools/gpu_session.sh
