#!/bin/bash
# round 6, session 1: where a lone 4K frame's decode / encode goes (kernel trace with start / end per launch), the two-stream
# overlap probe on the current kernels, and qoi_encode with the reference's worst-case allocation.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s1
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
R=$PWD
timeout 120 python tools/measure/single_trace.py 300 both 2>&1 | tail -1 | tee "$OUT/single_wall.txt"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/trace_dec" -o t -- python "$R/tools/measure/single_trace.py" 60 dec) > "$OUT/trace_dec.log" 2>&1
python tools/measure/trace_timeline.py "$OUT/trace_dec" "dec_transcode<0>" 40 | tee "$OUT/single_dec_timeline.txt"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/trace_enc" -o t -- python "$R/tools/measure/single_trace.py" 60 enc) > "$OUT/trace_enc.log" 2>&1
python tools/measure/trace_timeline.py "$OUT/trace_enc" "enc_sets" 40 | tee "$OUT/single_enc_timeline.txt"
rm -rf "$OUT/trace_dec" "$OUT/trace_enc"
echo "== overlap probe"
timeout 300 python tools/measure/overlap_probe.py 512 2 2>&1 | tail -1 | tee "$OUT/overlap.txt"
timeout 300 python tools/measure/overlap_probe.py 512 4 2>&1 | tail -1 | tee -a "$OUT/overlap.txt"
echo "== drop-in with the reference's worst-case buffer"
for W in 0 1; do
  QOIMI_ENCODE_WORST_CASE_BUFFER=$W timeout 300 python bench.py --frames 8 --steps 2 --warmup 1 --no-cpu --no-others --no-configs 2>/dev/null | tail -1 | \
    python -c "import json,sys; d=json.loads(sys.stdin.read()); print('worst_case=$W', json.dumps(d['dropin_host_pointers']))" | tee -a "$OUT/dropin_worst_case.txt"
done
echo "== done"
