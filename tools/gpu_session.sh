#!/bin/bash
# One gpurun call: probe + smoke + parity tests + bench + rocprof, each under its own timeout.
# Everything is logged under gpurun_out/ (merged back into the authoring container).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/s${SESSION:-1}
mkdir -p "$OUT"
export TMPDIR=/tmp
export PYTHONUNBUFFERED=1
echo "== env" | tee "$OUT/env.log"
(rocm-smi --showproductname 2>/dev/null | head -8; nproc; free -g | head -2) >> "$OUT/env.log" 2>&1

if [ "${DO_PROBE:-1}" = 1 ]; then
  echo "== lds probe"; timeout 120 ./build/lds_order_probe > "$OUT/probe.log" 2>&1; echo "rc=$?" >> "$OUT/probe.log"; tail -2 "$OUT/probe.log"
fi
if [ "${DO_SMOKE:-1}" = 1 ]; then
  echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "rc=$?" >> "$OUT/smoke.log"; tail -5 "$OUT/smoke.log"
fi
if [ "${DO_TESTS:-1}" = 1 ]; then
  echo "== pytest -m gpu"; timeout ${TEST_TIMEOUT:-1200} python -m pytest tests -m gpu -q --timeout 600 ${PYTEST_ARGS:-} > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -40 "$OUT/pytest.log"
fi
if [ "${DO_BENCH:-1}" = 1 ]; then
  echo "== bench"; timeout 600 python bench.py --steps ${BENCH_STEPS:-5} --warmup 2 ${BENCH_ARGS:-} > "$OUT/bench.log" 2>&1; echo "rc=$?" >> "$OUT/bench.log"; tail -5 "$OUT/bench.log"
fi
if [ "${DO_PROF:-1}" = 1 ]; then
  echo "== rocprofv3 kernel trace"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu ${BENCH_ARGS:-}) > "$OUT/prof.log" 2>&1
  echo "rc=$?" >> "$OUT/prof.log"; tail -3 "$OUT/prof.log"
  find "$OUT/prof" -name "*stats*" | head
fi
if [ -n "${EXTRA_CMD:-}" ]; then
  echo "== extra"; timeout ${EXTRA_TIMEOUT:-600} bash -c "$EXTRA_CMD" > "$OUT/extra.log" 2>&1; echo "rc=$?" >> "$OUT/extra.log"; tail -30 "$OUT/extra.log"
fi
echo "== done"
