#!/bin/bash
# per-kernel times of the step for the other content classes (256 frames)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; OUT=gpurun_out/${1:-s7}; mkdir -p $OUT; export TMPDIR=/tmp PYTHONUNBUFFERED=1
for kind in noise uiflat constant; do
  timeout 300 python bench.py --frames 256 --steps 3 --warmup 1 --no-cpu --no-others --no-single --no-configs --kind $kind 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['kernel_ms_per_step']; print('$kind', 'value', d['value'], 'exact', d['verified_bit_exact'], 'rounds', d['decode_rounds'], 'redo', d['decode_redo_segments'], 'syncfb', d['decode_sync_fallback_segments'], 'ms', d['ms_per_step'], {x: round(k[x],2) for x in k if k[x] > 0.05})
"
done | tee $OUT/kinds.txt
