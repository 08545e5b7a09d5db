#!/bin/bash
# environment-variable A/B inside one gpurun call (diagnostic): ENVS="a:X=1;b:X=0" KINDS="photo" BENCH_ARGS="--frames 64" bash tools/gpu_env_ab.sh name
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-ab}; mkdir -p $OUT; export TMPDIR=/tmp
IFS=';' read -ra VS <<< "${ENVS:-base:}"
for rep in 1 2; do
for v in "${VS[@]}"; do
  name=${v%%:*}; envs=${v#*:}
  for kind in ${KINDS:-photo}; do
    env $envs timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-others --no-single --kind $kind ${BENCH_ARGS:-} > $OUT/${name}_${kind}_$rep.log 2>&1
    python - $OUT/${name}_${kind}_$rep.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']; ok=True
        print(sys.argv[2], d['config']['content'], 'value', d['value'], 'exact', d['verified_bit_exact'], 'rounds', d.get('decode_rounds'), {x:k[x] for x in k if k[x]>0.05})
if not ok: print(sys.argv[2], 'FAILED'); print(open(sys.argv[1]).read()[-1500:])
PY
  done
done
done
