"""GPU diagnostic: per-kernel HIP-event times of a lone 4K frame (encode + decode), and wall clock per call."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from qoi_amd import api, synth
w, h = 3840, 2160
kind = sys.argv[1] if len(sys.argv) > 1 else "photo"
c = api.Context(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
npx = w * h; desc = api.QoiDesc(w, h, 4, 0)
ps = npx * 4; ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(ps, dtype=torch.uint8, device=dev); stt = torch.empty(ss, dtype=torch.uint8, device=dev)
out = torch.empty(ps, dtype=torch.uint8, device=dev); lens = torch.zeros(1, dtype=torch.int32, device=dev)
c.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 0, 1, w, h, px.data_ptr(), ps, st)
enc = lambda: c.encode_batch(px.data_ptr(), ps, desc, 1, stt.data_ptr(), ss, lens.data_ptr(), st)
enc(); c.encode_status(st); n = [int(lens[0].item())]
dec = lambda: c.decode_batch(stt.data_ptr(), ss, n, [desc], 4, out.data_ptr(), ps, st)
for _ in range(5): enc(); dec()
def wall(fn, reps=50):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
print(f"{kind}: wall encode {wall(enc):.4f} ms  decode {wall(dec):.4f} ms  both {wall(lambda: (enc(), dec())):.4f} ms; stats {c.decode_stats()}")
c.set_profiling(True)
for _ in range(20): enc(); dec()
prof = c.get_profile(st); c.set_profiling(False)
for k, v in prof.items():
    if v[1]: print(f"   {k:22s} {v[0] / 20 * 1e3:8.1f} us  ({v[1] // 20} marks per call)")
