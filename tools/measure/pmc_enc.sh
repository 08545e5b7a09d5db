#!/bin/bash
# SQ counters of the encode kernel alone (tools/measure/enc_time.py, 64 x 4K photographs) per arm of environment knobs:
# ARMS="name:ENV=..;name2:ENV=.." [EXTRA=1] bash tools/measure/pmc_enc.sh outdir      (LIB=<path from the repo root> in an arm picks another build)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-pmcenc}; mkdir -p $OUT; export TMPDIR=/tmp
IFS=';' read -ra AS <<< "${ARMS:-c0:}"
for a in "${AS[@]}"; do
  name=${a%%:*}; envs=${a#*:}; i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" \
             ${EXTRA:+"SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_ICACHE_BUSY_CYCLES SQ_INSTS_BRANCH"} \
             ${EXTRA:+"SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_INST_LEVEL_VMEM"}; do
    i=$((i+1))
    (cd /tmp && env $envs timeout 120 bash -c "exec rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OLDPWD/$OUT/${name}_$i -o pmc -- python $OLDPWD/tools/measure/enc_time.py \${LIB:--} ${FRAMES:-64}") > $OUT/${name}_$i.log 2>&1
    echo "$name pmc$i rc=$?"
  done
done
python - $OUT <<'PY'
import csv,glob,sys,collections,os
out=sys.argv[1]
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out+'/*/**/*counter_collection.csv', recursive=True):
    arm=f[len(out)+1:].split('/')[0].rsplit('_',1)[0]
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'].split('(')[0]
        if 'enc_sets' not in k: continue
        agg[(arm,k)][r['Counter_Name']].append(float(r['Counter_Value']))
with open(out+'/summary.txt','w') as fo:
    for key in sorted(agg):
        d={c: sum(v)/len(v) for c,v in agg[key].items()}
        w=d.get('SQ_WAVES',1)
        line=f"{key[0]:8s} {key[1][-30:]:30s} waves={w:.0f} " + ' '.join(f"{c.replace('SQ_','')}={d[c]/w:.4g}" for c in sorted(d) if c!='SQ_WAVES')
        print(line); fo.write(line+'\n')
PY
