#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of the decode kernels for build-flag variants, one gpurun call.  usage: VARIANTS="a:;b:-DX" bash tools/gpu_fetch.sh name
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-fetch}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-int-to-pointer-cast"
IFS=';' read -ra VS <<< "${VARIANTS:-base:}"
for v in "${VS[@]}"; do
  name=${v%%:*}; defs=${v#*:}
  ( cd qoi_amd/csrc && /opt/rocm/bin/hipcc $FLAGS $defs -c qoi_decode.hip -o ../lib/obj/qoi_decode.o && /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libqoi_mi355x.so ../lib/obj/*.o ) > $OUT/build_$name.log 2>&1 || { echo "build $name failed"; continue; }
  for set in FETCH_SIZE WRITE_SIZE; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OLDPWD/$OUT/${name}_$set -o pmc -- python $OLDPWD/bench.py --steps 2 --warmup 1 --no-cpu --no-others --no-single --no-configs --frames ${FRAMES:-256}) > $OUT/${name}_$set.log 2>&1
  done
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu --no-others --no-single --no-configs --frames ${FRAMES:-256} > $OUT/${name}_bench.log 2>&1
  python - $OUT $name <<'PY'
import csv,glob,sys,collections,json
out,name=sys.argv[1],sys.argv[2]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f'{out}/{name}_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','').replace('qoimi::','')
        if k.startswith('dec_'): agg[k][r['Counter_Name']].append((int(r['Grid_Size']),float(r['Counter_Value'])))
for l in open(f'{out}/{name}_bench.log'):
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']; print(name, 'exact', d['verified_bit_exact'], {x:k[x] for x in k if x.startswith('dec_') and k[x]>0.2})
for k in ('dec_transcode<0>','dec_summarize_rec<false>','dec_segments_rec<4>'):
    e=agg.get(k)
    if not e: continue
    s=[]
    for c in ('FETCH_SIZE','WRITE_SIZE'):
        v=e.get(c,[]); g=max(x[0] for x in v) if v else 0; vv=[x[1] for x in v if x[0]==g]
        s.append(f"{c} {sum(vv)/max(len(vv),1)*1024/1e9*(2 if c=='FETCH_SIZE' else 1):.2f} GB")
    print('   ',k,' '.join(s))
PY
done
