#!/bin/bash
# L2 <-> fabric request mix of the step's kernels (bench.py, 256 x 4K photo frames): how many write requests are full 64-byte ones, how many reads 32 / 64 / 128 bytes
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-pmctcc}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
i=0
for set in "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OLDPWD/$OUT/p$i -o pmc -- python $OLDPWD/bench.py --frames ${FRAMES:-256} --steps 2 --warmup 1 --no-cpu --no-others --no-single --no-configs) > $OUT/p$i.log 2>&1
  echo "pmc$i rc=$?"
done
python - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0].replace('void ','')
        if 'qoimi::' not in k: continue
        agg[k][r['Counter_Name']].append((int(r['Grid_Size']), float(r['Counter_Value'])))
with open(out+'/summary.txt','w') as fo:
    for k in sorted(agg):
        g=max(x[0] for v in agg[k].values() for x in v)
        d={c: (lambda vs: sum(vs)/len(vs))([x[1] for x in v if x[0]==g]) for c,v in agg[k].items()}
        if max(d.values()) < 1e6: continue
        line=f"{k[-42:]:42s} " + ' '.join(f"{c.replace('TCC_EA0_','').replace('_sum','')}={d[c]:.4g}" for c in sorted(d))
        print(line); fo.write(line+'\n')
PY
