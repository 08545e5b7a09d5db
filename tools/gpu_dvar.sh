#!/bin/bash
# decode build-flag variants inside ONE gpurun call: rebuilds qoi_decode.o only, relinks, runs the bench per variant.
# usage: VARIANTS="base:;arith:-DQOIMI_TR_LEN_ARITH" BENCH_ARGS="--frames 64" bash tools/gpu_dvar.sh name
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-dvar}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-int-to-pointer-cast"
FILE=${VFILE:-qoi_decode}
IFS=';' read -ra VS <<< "${VARIANTS:-base:}"
for v in "${VS[@]}"; do
  name=${v%%:*}; defs=${v#*:}
  ( cd qoi_amd/csrc && /opt/rocm/bin/hipcc $FLAGS $defs -c $FILE.hip -o ../lib/obj/$FILE.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libqoi_mi355x.so ../lib/obj/*.o ) > $OUT/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $OUT/build_$name.log; continue; }
  for kind in ${KINDS:-photo}; do
    env ${VENV:-} timeout 300 python bench.py --steps ${STEPS:-5} --warmup 2 --no-cpu --no-others --no-single --no-configs --frames ${FRAMES:-256} --kind $kind ${BENCH_ARGS:-} > $OUT/${name}_$kind.log 2>&1
    python - $OUT/${name}_$kind.log $name <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        ok=True; d=json.loads(l); k=d['kernel_ms_per_step']
        enc=k.get('encode_total') or sum(v for x,v in k.items() if x.startswith('enc_')); dec=k.get('decode_total') or sum(v for x,v in k.items() if x.startswith('dec_'))
        print(f"{sys.argv[2]:14s}", d['config']['content'], 'exact', d['verified_bit_exact'], 'enc', round(enc,3), 'dec', round(dec,3), {x:k[x] for x in k if k[x]>0.2})
if not ok: print(sys.argv[2], 'FAILED'); print(''.join(open(sys.argv[1]).readlines()[-8:]))
PY
  done
done
