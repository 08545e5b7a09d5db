#!/bin/bash
# One gpurun call: smoke + parity tests + bench (with CPU baseline) + rocprofv3 kernel stats + PMC traffic pass.
# Everything is logged under gpurun_out/$SESSION (merged back into the authoring container).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${SESSION:-sess}
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1
(rocm-smi --showproductname 2>/dev/null | head -8; nproc; free -g | head -2) > "$OUT/env.log" 2>&1
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "rc=$?" >> "$OUT/smoke.log"; tail -2 "$OUT/smoke.log"
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
echo "== bench"; timeout 900 python bench.py > "$OUT/bench.log" 2>&1; echo "rc=$?" >> "$OUT/bench.log"; tail -2 "$OUT/bench.log"
echo "== rocprofv3 kernel trace + stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu --no-single) > "$OUT/prof.log" 2>&1
echo "rc=$?" >> "$OUT/prof.log"
echo "== PMC: HBM traffic of the bench kernels (separate passes)"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OLDPWD/$OUT/pmc_$set" -o pmc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu --no-single) > "$OUT/pmc_$set.log" 2>&1
  echo "$set rc=$?"
done
python - "$OUT" <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'qoimi::' in k:
            agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k in sorted(agg):
    res[k] = {c: sum(v) / len(v) for c, v in agg[k].items()}
    res[k]['launches'] = max(len(v) for v in agg[k].values())
json.dump(res, open(out + '/pmc_traffic.json', 'w'), indent=1)
for k, v in res.items():
    print(k[:48], {c: round(x, 1) for c, x in v.items()})
PY
echo "== done"
