#!/bin/bash
# SQ counters of the decoder's kernels (bench.py, 256 x 4K photo frames), two passes: LDS side, issue side.  usage: bash tools/measure/pmc_dec.sh outdir
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-pmcdec}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" \
           "SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_VMEM"; do
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
        w=d.get('SQ_WAVES',1) or 1
        if w < 1000: continue
        line=f"{k[-42:]:42s} waves={w:.0f} " + ' '.join(f"{c.replace('SQ_','')}={d[c]/w:.4g}" for c in sorted(d) if c!='SQ_WAVES')
        print(line); fo.write(line+'\n')
PY
