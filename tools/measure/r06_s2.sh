#!/bin/bash
# round 6, session 2: dec_scan_entry (single-pass look-back for pixel offsets + speculated slots) on lone frames: bytes, timeline.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${SESSION:-r06_s2}
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
R=$PWD
for K in photo noise uiflat; do
  for F in 1 0; do KIND=$K QOIMI_DEC_FUSED=$F timeout 120 python tools/measure/single_trace.py 300 dec 2>&1 | tail -1 | sed "s/^/$K fused=$F /"; done
done | tee "$OUT/single_wall.txt"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/trace_dec" -o t -- python "$R/tools/measure/single_trace.py" 60 dec) > "$OUT/trace_dec.log" 2>&1
python tools/measure/trace_timeline.py "$OUT/trace_dec" "dec_transcode<0" 40 | head -24 | tee "$OUT/single_dec_timeline.txt"
rm -rf "$OUT/trace_dec"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --timeout 600 -k "${TESTS:-4k_frame or decode or golden or hostile or small or segment or flat or record or selectable or shapes}" > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -5 "$OUT/pytest.log"
echo "== done"
