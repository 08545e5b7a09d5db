#!/bin/bash
# PMC passes (separate runs, kernel-trace only) for the bench kernels -> gpurun_out/$1/pmc_summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-pmc}; mkdir -p $OUT; export TMPDIR=/tmp
P="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu --no-others ${BENCH_ARGS:-}"
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
           "SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES_EQ_64 SQ_INST_LEVEL_VMEM" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OLDPWD/$OUT/pmc$i -o pmc -- $P) > $OUT/pmc$i.log 2>&1
  echo "pmc$i rc=$?"
done
python - $OUT <<'PY'
import csv,glob,sys,collections
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/pmc*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'qoimi' not in k: continue
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+'/pmc_summary.txt','w') as fo:
    for k in sorted(agg):
        fo.write(k+'\n')
        for c in sorted(agg[k]):
            v=agg[k][c]; fo.write(f'   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}\n')
keys=['SQ_WAVES','SQ_WAVE_CYCLES','SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_INSTS_VMEM_RD','SQ_INSTS_VMEM_WR','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','GRBM_GUI_ACTIVE','FETCH_SIZE','WRITE_SIZE','TCC_HIT_sum','TCC_MISS_sum']
import os
flt=os.environ.get('PMC_FILTER','')
for k in sorted(agg):
    if flt and flt not in k: continue
    print(k[:60]); print('   '+'  '.join(f"{c.replace('SQ_','')}={sum(agg[k][c])/len(agg[k][c]):.3g}" for c in keys if c in agg[k]))
PY
