#!/bin/bash
# quick GPU iteration: parity tests + encode-only / full bench with per-kernel times -> gpurun_out/$1
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-q}; mkdir -p $OUT; export TMPDIR=/tmp
if [ "${DO_TESTS:-1}" = 1 ]; then
  timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -15 $OUT/pytest.log
fi
for kind in ${KINDS:-photo}; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-others --kind $kind ${BENCH_ARGS:-} > $OUT/bench_$kind.log 2>&1; echo "rc=$?" >> $OUT/bench_$kind.log
  python - $OUT/bench_$kind.log <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(d['config']['content'], 'value', d['value'], 'exact', d['verified_bit_exact'], 'enc', d['encode_mpps_kernels'], 'dec', d['decode_mpps_kernels'], 'frac', d['roofline']['frac']); print(d['kernel_ms_per_step']); print(d.get('single_frame'))
    elif 'rror' in l or 'rc=' in l: print(l.strip())
PY
done
if [ -n "${EXTRA_CMD:-}" ]; then timeout ${EXTRA_TIMEOUT:-600} bash -c "$EXTRA_CMD" > $OUT/extra.log 2>&1; echo "rc=$?" >> $OUT/extra.log; tail -30 $OUT/extra.log; fi
