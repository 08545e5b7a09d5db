#!/bin/bash
# round 6, session 32: the batch model of choose_seg_bytes: mixed directory, batches of 8 .. 256 frames by class, the alternating / headline legs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r06_s32
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0 QOIMI_TUNING=1
ulimit -c 0
for V in "QOIMI_SEG_BYTES=" "QOIMI_SEG_BYTES=256" "QOIMI_SEG_BYTES=512" "QOIMI_SEG_BYTES=1024"; do
  env $V timeout 300 python tools/measure/mixed_trace.py 2>&1 | tail -2 | sed "s/^/$V /"
done | cut -c1-1300 | tee "$OUT/mixed_by_seg.txt"
for K in photo sprite_alpha photo_hard noise uiflat; do for F in 8 32 128 256; do
  KIND=$K timeout 200 python tools/measure/dec_time.py - $F 2>&1 | tail -1 | sed -E "s/^.*'decode_total'/total/" | sed "s/^/$K F=$F /"
done; done | tee "$OUT/batch_auto.txt"
timeout 900 python bench.py --no-cpu > "$OUT/bench.json" 2> "$OUT/bench.err"; tail -c 600 "$OUT/bench.err"
python - <<'PY' | tee "$OUT/bench_brief.txt"
import json
d = json.loads(open("gpurun_out/r06_s32/bench.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms_per_step", d["ms_per_step"], "kernel", d["kernel_ms_per_step"])
print("single", {k: d["single_frame"][k] for k in ("ms", "encode_ms", "decode_ms")})
print("mixed", {k: d["mixed_directory"][k] for k in ("encode_ms", "decode_ms", "decode_rounds", "verified_bit_exact")})
print("alternating", d["alternating"]["per_class"])
print("others", {k: (v["encode_ms"], v["decode_ms"], v["decode_rounds"]) for k, v in d["other_content"].items()})
PY
