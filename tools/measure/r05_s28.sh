#!/bin/bash
# round 5, session 28: bench.py's content classes again (photo_hard read 56.9 ms of decode in session 24, 45-47 ms in every measurement of its own).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r05_s28
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu --no-single --no-configs > "$OUT/bench.log" 2>&1; echo "rc=$?" >> "$OUT/bench.log"
python - "$OUT/bench.log" <<'PY' | tee "$OUT/others.txt"
import json, sys
line = [l for l in open(sys.argv[1]).read().splitlines() if l.lstrip().startswith('{"metric"')][0]
b = json.loads(line)
print("headline", b["value"], b["ms_per_step"], b["kernel_ms_per_step"].get("encode_total"), b["kernel_ms_per_step"].get("decode_total"))
for k, v in b["other_content"].items():
    print(k, v["ms_per_step"], v["encode_ms"], v["decode_ms"], v["roofline_encode_frac"], v["roofline_decode_frac"], v["decode_rounds"])
PY
echo "== done"
