#!/bin/bash
# round 6, session 4: what the strided per-segment state I/O costs P3 / P4 on a lone 4K frame (timing-only builds: [slot][segment] addressing)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s4
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1 QOIMI_DEC_MAX_ROUNDS=1
ulimit -c 0
R=$PWD
for L in qoi_amd/lib build/exp_p4x; do
  (cd /tmp && QOIMI_TOOLS_LIB=$R/$L/libqoi_mi355x.so timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/trace" -o t -- python "$R/tools/measure/single_trace.py" 60 dec) > "$OUT/trace.log" 2>&1
  python tools/measure/trace_timeline.py "$OUT/trace" "dec_transcode<0" 40 | grep -E "dec_summarize|dec_segments|chain_state" | sed "s|^|$L |"
  rm -rf "$OUT/trace"
done | tee "$OUT/state_io.txt"
