"""Decode-only timing of one library build: python tools/measure/dec_time.py <libqoi_mi355x.so | -> [frames] - per-kernel ms of
qoimi_decode_batch on the streams of 4K frames (KIND=photo|noise|uiflat|constant); says whether the pixels came back right, does
not insist (for builds whose pixels are wrong on purpose: run those with QOIMI_DEC_MAX_ROUNDS=1)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qoi_amd import api, synth
if len(sys.argv) > 1 and sys.argv[1] != '-':
    api.LIB_PATH = os.path.abspath(sys.argv[1])
F = int(sys.argv[2]) if len(sys.argv) > 2 else 256
w, h = 3840, 2160
c = api.Context(0)
npx = w * h
ps = npx * 4
ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(F * ps, dtype=torch.uint8, device='cuda'); st = torch.empty(F * ss, dtype=torch.uint8, device='cuda')
out = torch.empty(F * ps, dtype=torch.uint8, device='cuda'); lens = torch.zeros(F, dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream().cuda_stream
c.synth_frames(synth.KIND_ID[os.environ.get("KIND", "photo")], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, s)
desc = api.QoiDesc(w, h, 4, 0)
c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s); c.encode_status(s)
sizes = [int(v) for v in lens.cpu().tolist()]
reps = int(os.environ.get("REPS", "3"))
c.decode_batch(st.data_ptr(), ss, sizes, [desc] * F, 4, out.data_ptr(), ps, s)
c.set_profiling(True)
for _ in range(reps):
    c.decode_batch(st.data_ptr(), ss, sizes, [desc] * F, 4, out.data_ptr(), ps, s)
prof = c.get_profile(s)
c.set_profiling(False)
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    c.decode_batch(st.data_ptr(), ss, sizes, [desc] * F, 4, out.data_ptr(), ps, s)
torch.cuda.synchronize(); wall_ms = (time.perf_counter() - t0) / reps * 1e3
print(os.path.basename(os.path.dirname(api.LIB_PATH)), {k: round(v[0] / reps, 3) for k, v in prof.items() if v[1] and v[0] / reps > 0.02},
      'rounds', c.decode_stats()['rounds'], 'equal', bool(torch.equal(out, px)), 'wall ms', round(wall_ms, 2), 'workspace GB', round(c.workspace_bytes()['decode'] / 1e9, 2), 'streams GB', round(sum(sizes) / 1e9, 2))
