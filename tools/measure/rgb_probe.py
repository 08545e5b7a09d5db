import os, sys, time
sys.path.insert(0, '/root/repo')
import torch
from qoi_amd import api, synth
F = 256; w, h = 3840, 2160; npx = w*h
c = api.Context(0)
ps4 = npx*4; ps3 = (npx*3 + 255)//256*256; ss = (api.encode_bound(w, h, 4) + 255)//256*256
px = torch.empty(F*ps4, dtype=torch.uint8, device='cuda'); p3 = torch.empty(F*ps3, dtype=torch.uint8, device='cuda'); st = torch.empty(F*ss, dtype=torch.uint8, device='cuda'); lens = torch.zeros(F, dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream().cuda_stream
c.synth_frames(synth.KIND_ID['photo'], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps4, s)
v3 = p3.view(F, ps3)
for lo in range(0, F, 16):
    v3[lo:lo+16, :npx*3] = px[lo*ps4:(lo+16)*ps4].view(16, npx, 4)[:, :, :3].reshape(16, npx*3)
d3 = api.QoiDesc(w, h, 3, 0); d4 = api.QoiDesc(w, h, 4, 0)
def t(fn, n=5):
    fn(); fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
e3 = lambda: c.encode_batch(p3.data_ptr(), ps3, d3, F, st.data_ptr(), ss, lens.data_ptr(), s)
e4 = lambda: c.encode_batch(px.data_ptr(), ps4, d4, F, st.data_ptr(), ss, lens.data_ptr(), s)
print(dict((k, os.environ[k]) for k in os.environ if k.startswith('QOIMI_')), 'rgb encode ms', round(t(e3), 3), 'rgba encode ms', round(t(e4), 3))
