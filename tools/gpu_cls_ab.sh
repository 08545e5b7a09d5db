#!/bin/bash
# ONE gpurun call for the matrix-pipe classes of the encoder (QOIMI_ENC_CLS): the whole GPU suite with CLS 1 forced for every
# context, the bench per arm (per-kernel times, reference check of the bench's own frames), the encode tests with CLS 2, and a
# rocprofv3 kernel trace of the CLS 1 arm.  Everything lands in gpurun_out/$1 as it goes.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; NAME=${1:-cls}; OUT=gpurun_out/$NAME; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
echo "== pytest, QOIMI_ENC_CLS=1 for every context"
QOIMI_ENC_CLS=1 timeout 420 python -m pytest tests -m gpu -x -q --timeout 300 > $OUT/pytest_cls1.log 2>&1; echo "rc=$?" >> $OUT/pytest_cls1.log; tail -4 $OUT/pytest_cls1.log
echo "== bench arms"
DO_TESTS=0 FRAMES=${FRAMES:-256} ARMS="${ARMS:-c0:;c1:QOIMI_ENC_CLS=1;c2:QOIMI_ENC_CLS=2;c2r4:QOIMI_ENC_CLS=2 QOIMI_ENC_SET_SLABS=4}" bash tools/gpu_enc_iter.sh $NAME
echo "== encode tests, QOIMI_ENC_CLS=2"
QOIMI_ENC_CLS=2 timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q --timeout 300 -k "encode or sweep or selectable or mixed or flat_frames or set_sizes or one_context or three_channel or 4k_frame or batch_1080p" > $OUT/pytest_cls2.log 2>&1; echo "rc=$?" >> $OUT/pytest_cls2.log; tail -3 $OUT/pytest_cls2.log
echo "== rocprofv3 kernel trace, CLS 1"
(cd /tmp && QOIMI_ENC_CLS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof_cls1" -o trace -- python "$OLDPWD/bench.py" --frames ${FRAMES:-256} --steps 5 --warmup 2 --no-cpu --no-others --no-single --no-configs) > $OUT/prof_cls1.log 2>&1; echo "rc=$?" >> $OUT/prof_cls1.log
f=$(ls $OUT/prof_cls1/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -12 "$f"
echo "== done"
