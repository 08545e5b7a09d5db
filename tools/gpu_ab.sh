#!/bin/bash
# A/B of environment knobs inside ONE gpurun call (boxes differ by several per cent): parity tests once, then the bench per arm.
# usage: ARMS="b2k:QOIMI_SEG_BYTES=2048;b1k:QOIMI_SEG_BYTES=1024" BENCH_ARGS="--frames 64" KINDS="photo" bash tools/gpu_ab.sh name
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-ab}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
if [ "${DO_TESTS:-1}" = 1 ]; then
  timeout 900 python -m pytest tests -m gpu -x -q --timeout 600 ${PYTEST_ARGS:-} > $OUT/pytest.log 2>&1; echo "rc=$?" >> $OUT/pytest.log; tail -12 $OUT/pytest.log
fi
IFS=';' read -ra AS <<< "${ARMS:-base:}"
for rep in $(seq 1 ${REPS:-1}); do
for a in "${AS[@]}"; do
  name=${a%%:*}; envs=${a#*:}
  for kind in ${KINDS:-photo}; do
    env $envs timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu --no-others --no-configs --frames ${FRAMES:-256} --kind $kind ${BENCH_ARGS:-} > $OUT/${name}_${kind}_$rep.log 2>&1; echo "rc=$?" >> $OUT/${name}_${kind}_$rep.log
    python - $OUT/${name}_${kind}_$rep.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        ok=True
        d=json.loads(l); k=d['kernel_ms_per_step']
        enc=k.get('encode_total') or sum(v for x,v in k.items() if x.startswith('enc_')); dec=k.get('decode_total') or sum(v for x,v in k.items() if x.startswith('dec_'))
        print(sys.argv[2], d['config']['content'], 'value', d['value'], 'ms', d['ms_per_step'], 'exact', d['verified_bit_exact'], 'rounds', d.get('decode_rounds'), 'enc_ms', round(enc,3), 'dec_ms', round(dec,3), 'single', (d.get('single_frame') or {}).get('ms'))
        print('   ', {x:k[x] for x in k if k[x]>0.02})
if not ok:
    print(sys.argv[2], 'FAILED'); print(''.join(open(sys.argv[1]).readlines()[-12:]))
PY
  done
done
done
if [ -n "${EXTRA_CMD:-}" ]; then timeout ${EXTRA_TIMEOUT:-600} bash -c "$EXTRA_CMD" > $OUT/extra.log 2>&1; echo "rc=$?" >> $OUT/extra.log; tail -30 $OUT/extra.log; fi
