"""GPU diagnostic: decode kernels with 3-channel output beside 4-channel output (64 4K photo frames)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from qoi_amd import api, synth
w, h, F = 3840, 2160, 64
c = api.Context(0); dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
npx = w * h; desc = api.QoiDesc(w, h, 4, 0)
ps = npx * 4; ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(F * ps, dtype=torch.uint8, device=dev); stt = torch.empty(F * ss, dtype=torch.uint8, device=dev)
out = torch.empty(F * ps, dtype=torch.uint8, device=dev); lens = torch.zeros(F, dtype=torch.int32, device=dev)
c.synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, st)
c.encode_batch(px.data_ptr(), ps, desc, F, stt.data_ptr(), ss, lens.data_ptr(), st); c.encode_status(st)
sizes = [int(x) for x in lens.cpu().numpy()]
for och in (4, 3):
    stride = npx * och
    for _ in range(2): c.decode_batch(stt.data_ptr(), ss, sizes, [desc] * F, och, out.data_ptr(), stride, st)
    c.set_profiling(True)
    for _ in range(3): c.decode_batch(stt.data_ptr(), ss, sizes, [desc] * F, och, out.data_ptr(), stride, st)
    prof = c.get_profile(st); c.set_profiling(False)
    ref = px.view(F, npx, 4)[:, :, :och].reshape(F, -1)
    ok = bool(torch.equal(out[:F * stride].view(F, stride), ref))
    print("channels", och, "exact", ok, {k: round(v[0] / 3, 3) for k, v in prof.items() if v[1] and v[0] / 3 > 0.05})
