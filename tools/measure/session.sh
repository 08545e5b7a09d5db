#!/bin/bash
# One gpurun call: smoke + parity tests + bench (with CPU baseline) + rocprofv3 kernel stats + PMC traffic pass.
# Everything is logged under gpurun_out/$SESSION (merged back into the authoring container).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${SESSION:-sess}
mkdir -p "$OUT"
export TMPDIR=/tmp PYTHONUNBUFFERED=1 HSA_ENABLE_COREDUMP=0
ulimit -c 0
(rocm-smi --showproductname 2>/dev/null | head -8; nproc; free -g | head -2) > "$OUT/env.log" 2>&1
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; echo "rc=$?" >> "$OUT/smoke.log"; tail -2 "$OUT/smoke.log"
echo "== pytest -m gpu"; timeout 1200 python -m pytest tests -m gpu -q --timeout 600 > "$OUT/pytest.log" 2>&1; echo "rc=$?" >> "$OUT/pytest.log"; tail -4 "$OUT/pytest.log"
echo "== bench"; timeout 900 python bench.py > "$OUT/bench.log" 2>&1; echo "rc=$?" >> "$OUT/bench.log"; tail -2 "$OUT/bench.log"
echo "== rocprofv3 kernel trace + stats"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o trace -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu --no-others --no-single --no-configs) > "$OUT/prof.log" 2>&1
echo "rc=$?" >> "$OUT/prof.log"
[ "${SKIP_PMC:-0}" = 1 ] && { echo "== done (PMC passes skipped)"; exit 0; }
echo "== PMC: HBM traffic of the bench kernels (separate passes)"
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$OLDPWD/$OUT/pmc_$set" -o pmc -- python "$OLDPWD/bench.py" --steps 3 --warmup 1 --no-cpu --no-others --no-single --no-configs) > "$OUT/pmc_$set.log" 2>&1
  echo "$set rc=$?"
done
python - "$OUT" <<'PY'
# per kernel: mean FETCH_SIZE / WRITE_SIZE (KB, as reported) over the batch launches = the launches with the
# kernel's largest grid (verification and single-image launches use smaller grids)
import csv, glob, json, sys, collections
out = sys.argv[1]
rows = collections.defaultdict(list)
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        if 'qoimi::' in k:
            rows[k].append((int(r['Grid_Size']), r['Counter_Name'], float(r['Counter_Value'])))
kern = {}
for k in sorted(rows):
    g = max(x[0] for x in rows[k])
    e = {'grid': g}
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        v = [x[2] for x in rows[k] if x[0] == g and x[1] == c]
        if v:
            e[c + '_KB'] = round(sum(v) / len(v), 1); e['launches_averaged'] = len(v)
    kern[k] = e
doc = {"command": "rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu --no-others --no-single --no-configs (two separate passes, tools/measure/session.sh)",
       "frames": 1024, "commit": (open('build/HEAD').read().strip() if __import__('os').path.exists('build/HEAD') else 'not recorded'),
       "note": "per-launch means over the batch launches (largest grid of each kernel); KB as rocprofv3 reports them. gfx950: FETCH_SIZE tallies 128-byte read requests at 64 bytes (MI355X_MICROARCH.md HBM section); calibration: enc_slab_summary is a pure streaming read of every pixel byte of the batch, its FETCH_SIZE comes out at ~0.5 x those bytes -> read bytes = 2 x FETCH_SIZE. WRITE_SIZE is taken as reported.",
       "read_correction": 2.0, "kernels": kern}
json.dump(doc, open(out + '/pmc_traffic.json', 'w'), indent=1)
for k, v in kern.items():
    print(k[:56], v)
PY
echo "== done"
