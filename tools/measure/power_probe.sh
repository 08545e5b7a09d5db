#!/bin/bash
# power / clock of the GPU while one kernel family runs in a loop: tools/measure/power_probe.sh <enc|dec> [lib]
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export PYTHONUNBUFFERED=1
what=$1; lib=${2:-qoi_amd/lib/libqoi_mi355x.so}
rocm-smi --showmaxpower --showpower --showclocks 2>/dev/null | grep -v "^=\|^$" | head -20
python - "$what" "$lib" <<'PY' &
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from qoi_amd import api, synth
api.LIB_PATH = os.path.abspath(sys.argv[2])
F = 512; w, h = 3840, 2160
c = api.Context(0); npx = w * h; ps = npx * 4; ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(F * ps, dtype=torch.uint8, device='cuda'); st = torch.empty(F * ss, dtype=torch.uint8, device='cuda')
out = torch.empty(F * ps, dtype=torch.uint8, device='cuda'); lens = torch.zeros(F, dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream().cuda_stream
c.synth_frames(synth.KIND_ID[os.environ.get("KIND", "photo")], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, s)
desc = api.QoiDesc(w, h, 4, 0)
c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s); c.encode_status(s)
sizes = [int(v) for v in lens.cpu().tolist()]
t0 = time.time(); n = 0
print("loop start", flush=True)
while time.time() - t0 < 12:
    if sys.argv[1] == 'enc':
        for _ in range(20): c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s)
        c.encode_status(s)
    else:
        for _ in range(5): c.decode_batch(st.data_ptr(), ss, sizes, [desc] * F, 4, out.data_ptr(), ps, s)
    n += 1
print("loop end", n, flush=True)
PY
sleep 6
for i in 1 2 3 4 5 6; do rocm-smi --showpower --showclocks 2>/dev/null | grep -i "power\|sclk\|mclk\|fclk" | tr '\n' ' '; echo; sleep 1.5; done
wait
