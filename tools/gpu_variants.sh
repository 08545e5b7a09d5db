#!/bin/bash
# build-flag ablations on the GPU box (diagnostic): rebuilds qoi_decode.hip / qoi_encode.hip with extra -D flags and runs the bench
# usage: VARIANTS="base:;g4:-DQOIMI_DRAIN_GROUP=4" KINDS="photo" BENCH_ARGS="--frames 32" bash tools/gpu_variants.sh name
# a variant "name:flags:file" builds with qoi_amd/csrc/<file> in place of qoi_decode.hip (an earlier revision kept beside
# it for the comparison, e.g. `git show HEAD~1:qoi_amd/csrc/qoi_decode.hip > qoi_amd/csrc/_prev_decode.hip`): variants are
# only comparable inside one call, the boxes differ by several per cent
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-var}; mkdir -p $OUT; export TMPDIR=/tmp
BASEFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-int-to-pointer-cast"
IFS=';' read -ra VS <<< "${VARIANTS:-base:}"
for v in "${VS[@]}"; do
  name=${v%%:*}; rest=${v#*:}; defs=${rest%%:*}; alt=""; [ "$rest" != "$defs" ] && alt=${rest#*:}
  [ -f qoi_amd/csrc/.cur_decode.hip ] || cp qoi_amd/csrc/qoi_decode.hip qoi_amd/csrc/.cur_decode.hip
  if [ -n "$alt" ]; then cp qoi_amd/csrc/$alt qoi_amd/csrc/qoi_decode.hip; else cp qoi_amd/csrc/.cur_decode.hip qoi_amd/csrc/qoi_decode.hip; fi
  touch qoi_amd/csrc/qoi_decode.hip qoi_amd/csrc/qoi_encode.hip
  make -C qoi_amd/csrc FLAGS="$BASEFLAGS $defs" > $OUT/build_$name.log 2>&1 || { echo "build $name failed"; tail -5 $OUT/build_$name.log; continue; }
  for kind in ${KINDS:-photo}; do
    timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-others --no-single --kind $kind ${BENCH_ARGS:-} > $OUT/${name}_$kind.log 2>&1
    python - $OUT/${name}_$kind.log $name <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']
        print(sys.argv[2], d['config']['content'], 'value', d['value'], 'exact', d['verified_bit_exact'], 'rounds', d.get('decode_rounds'), {x:k[x] for x in k if k[x]>0.05})
PY
  done
done
cp qoi_amd/csrc/.cur_decode.hip qoi_amd/csrc/qoi_decode.hip; rm -f qoi_amd/csrc/.cur_decode.hip
