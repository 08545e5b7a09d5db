#!/bin/bash
# encoder iteration inside ONE gpurun call: encode-related parity tests first, then the bench (encode + decode, per-kernel times)
# per arm of environment knobs.  usage: ARMS="lb:;of:QOIMI_ENC_LOOKBACK=0" FRAMES=256 bash tools/gpu_enc_iter.sh name
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-enc}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
if [ "${DO_TESTS:-1}" = 1 ]; then
  timeout ${TEST_TIMEOUT:-900} python -m pytest tests -m gpu -x -q --timeout 600 ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -${TEST_TAIL:-15} $OUT/pytest.log
fi
IFS=';' read -ra AS <<< "${ARMS:-base:}"
for a in "${AS[@]}"; do
  name=${a%%:*}; envs=${a#*:}
  for kind in ${KINDS:-photo}; do
    env $envs timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu --no-others --no-single --no-configs --frames ${FRAMES:-256} --kind $kind ${BENCH_ARGS:-} > $OUT/${name}_${kind}.log 2>&1; echo "rc=$?" >> $OUT/${name}_${kind}.log
    python - $OUT/${name}_${kind}.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        ok=True
        d=json.loads(l); k=d['kernel_ms_per_step']
        print(sys.argv[2], d['config']['content'], 'value', d['value'], 'exact', d['verified_bit_exact'], d.get('reference_check'), 'enc_ms', k.get('encode_total'), 'dec_ms', k.get('decode_total'), 'frac', d['roofline']['frac'], 'B/px', d['config'].get('stream_bytes_per_px'))
        print('   ', {x:round(k[x],3) for x in k if k[x]>0.02})
if not ok:
    print(sys.argv[2], 'FAILED'); print(''.join(open(sys.argv[1]).readlines()[-12:]))
PY
  done
done
if [ -n "${EXTRA_CMD:-}" ]; then timeout ${EXTRA_TIMEOUT:-600} bash -c "$EXTRA_CMD" > $OUT/extra.log 2>&1; echo "rc=$?" >> $OUT/extra.log; tail -30 $OUT/extra.log; fi
