#!/bin/bash
# round 5, session 32: bench.py's class lines with and without the all-by-state-look-back form (same box).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s32
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
for A in 1 0; do
QOIMI_ENC_ALL_G2=$A timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-single --no-configs > "$OUT/bench_$A.log" 2>&1; echo "rc=$?" >> "$OUT/bench_$A.log"
python - "$OUT/bench_$A.log" $A <<'PY' | tee -a "$OUT/others.txt"
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.lstrip().startswith('{"metric"')][0]
b = json.loads(line)
print("all_g2=" + sys.argv[2], "headline", b["value"], b["ms_per_step"])
for k, v in b["other_content"].items():
    print("all_g2=" + sys.argv[2], k, v["ms_per_step"], v["encode_ms"], v["decode_ms"], v["roofline_encode_frac"], v["roofline_decode_frac"])
PY
done
echo "== done"
