"""Do encode and decode overlap when two host threads drive two contexts on two streams?  Each thread round-trips its part of F 4K
photo frames; compared with one thread doing all F.  python tools/measure/overlap_probe.py [F] [parts]"""
import os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qoi_amd import api, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 512
P = int(sys.argv[2]) if len(sys.argv) > 2 else 2
w, h = 3840, 2160
npx = w * h; ps = npx * 4; ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(F * ps, dtype=torch.uint8, device='cuda'); st = torch.empty(F * ss, dtype=torch.uint8, device='cuda')
out = torch.empty(F * ps, dtype=torch.uint8, device='cuda'); lens = torch.zeros(F, dtype=torch.int32, device='cuda')
desc = api.QoiDesc(w, h, 4, 0)
c0 = api.Context(0)
s0 = torch.cuda.current_stream().cuda_stream
c0.synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, s0)
torch.cuda.synchronize()

def trip(c, s, lo, n, stagger=0):
    c.encode_batch(px.data_ptr() + lo * ps, ps, desc, n, st.data_ptr() + lo * ss, ss, lens.data_ptr() + lo * 4, s); c.encode_status(s)
    sizes = [int(v) for v in lens[lo:lo + n].cpu().tolist()]
    c.decode_batch(st.data_ptr() + lo * ss, ss, sizes, [desc] * n, 4, out.data_ptr() + lo * ps, ps, s)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

one = timed(lambda: trip(c0, s0, 0, F))
ctxs = [api.Context(0) for _ in range(P)]
streams = [torch.cuda.Stream() for _ in range(P)]
def many():
    th = []
    for i in range(P):
        n = F // P
        t = threading.Thread(target=trip, args=(ctxs[i], streams[i].cuda_stream, i * n, n)); t.start(); th.append(t)
    for t in th: t.join()
par = timed(many)
# sequential parts on one context: what the split alone costs
seq = timed(lambda: [trip(c0, s0, i * (F // P), F // P) for i in range(P)])
print(f"F={F}: one call {one:.2f} ms, {P} parts in sequence {seq:.2f} ms, {P} threads / contexts / streams {par:.2f} ms, equal {bool(torch.equal(out, px))}")
