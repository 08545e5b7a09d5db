import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from qoi_amd import api, synth
def go(kind, F, w, h, plain, seg=None):
    os.environ["QOIMI_P3_PLAIN"] = plain
    if seg: os.environ["QOIMI_SEG_BYTES"] = str(seg)
    else: os.environ.pop("QOIMI_SEG_BYTES", None)
    c = api.Context(0); dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    npx = w * h; desc = api.QoiDesc(w, h, 4, 0)
    ps = (npx * 4 + 255) // 256 * 256; ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
    px = torch.empty(F * ps, dtype=torch.uint8, device=dev); stt = torch.empty(F * ss, dtype=torch.uint8, device=dev)
    out = torch.zeros(F * ps, dtype=torch.uint8, device=dev); lens = torch.zeros(F, dtype=torch.int32, device=dev)
    c.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, st)
    c.encode_batch(px.data_ptr(), ps, desc, F, stt.data_ptr(), ss, lens.data_ptr(), st); c.encode_status(st)
    sizes = [int(x) for x in lens.cpu().numpy()]
    c.decode_batch(stt.data_ptr(), ss, sizes, [desc] * F, 4, out.data_ptr(), ps, st); torch.cuda.synchronize()
    print(kind, F, w, h, "seg", seg, "plain", plain, c.decode_stats(), "exact", bool(torch.equal(out.view(F, ps)[:, :npx*4], px.view(F, ps)[:, :npx*4])), flush=True)
for plain in ("1", "0"):
    go("photo", 1, 64, 64, plain, 128); go("photo", 1, 512, 512, plain, 128); go("photo", 1, 512, 512, plain, 2048); go("photo", 2, 3840, 2160, plain); go("noise", 1, 512, 512, plain, 128)
