#!/bin/bash
# encode ablations + PMC passes (diagnostics; outputs under gpurun_out/$1)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-abl}; mkdir -p $OUT
export TMPDIR=/tmp
B="python bench.py --steps 5 --warmup 1 --no-cpu --no-others --encode-only"
for v in "base:" "abl1:QOIMI_ENC_ABLATE=1" "abl2:QOIMI_ENC_ABLATE=2" "abl4:QOIMI_ENC_ABLATE=4" "abl7:QOIMI_ENC_ABLATE=7" "noticket:QOIMI_ENC_TICKET=0" "probe0:QOIMI_ENC_PROBE=0"; do
  name=${v%%:*}; envs=${v#*:}
  echo "== $name"; env $envs timeout 300 $B > $OUT/$name.log 2>&1
  python - "$OUT/$name.log" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print({k:v for k,v in d['kernel_ms_per_step'].items() if k.startswith('enc')}, d['roofline']['frac'])
PY
done
for kind in noise uiflat constant; do
  echo "== kind $kind"; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu --no-others --kind $kind > $OUT/kind_$kind.log 2>&1
  python - "$OUT/kind_$kind.log" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['verified_bit_exact'], d['decode_rounds'], d['kernel_ms_per_step'])
PY
done
if [ "${DO_PMC:-1}" = 1 ]; then
  rocprofv3 -L > $OUT/counters.txt 2>&1
  P="python $PWD/bench.py --steps 2 --warmup 1 --no-cpu --no-others"
  i=0
  for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
             "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA" \
             "FETCH_SIZE" "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OLDPWD/$OUT/pmc$i -o pmc -- $P) > $OUT/pmc$i.log 2>&1
    echo "pmc$i rc=$?"
  done
  ls $OUT/pmc1 2>/dev/null | head
fi
