"""Encode-only timing of one library build: python tools/measure/enc_time.py <libqoi_mi355x.so> [frames] - per-kernel ms of
qoimi_encode_batch on 4K photo frames (no checks: for builds whose bytes are wrong on purpose)."""
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
px = torch.empty(F * ps, dtype=torch.uint8, device='cuda'); st = torch.empty(F * ss, dtype=torch.uint8, device='cuda'); lens = torch.zeros(F, dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream().cuda_stream
c.synth_frames(synth.KIND_ID[os.environ.get("KIND", "photo")], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, s)
desc = api.QoiDesc(w, h, 4, 0)
for _ in range(2):
    c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s); c.encode_status(s)
c.set_profiling(True)
for _ in range(5):
    c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s); c.encode_status(s)
prof = c.get_profile(s)
c.set_profiling(False)
import time
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3):
    c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s)
torch.cuda.synchronize(); wall_ms = (time.perf_counter() - t0) / 3 * 1e3
print(os.environ.get('QOIMI_ENC_SET_SLABS', '-'), os.path.basename(os.path.dirname(api.LIB_PATH)), {k: round(v[0] / 5, 3) for k, v in prof.items() if v[1] and v[0] / 5 > 0.02}, 'wall ms', round(wall_ms, 3), 'bytes/px', round(float(lens.sum()) / (F * npx), 4))
