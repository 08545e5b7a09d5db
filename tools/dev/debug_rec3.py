"""GPU diagnostic: uiflat frames on the record pipeline, output prefilled with 0xAB to tell unwritten from wrong pixels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from qoi_amd import api, synth
w, h = 3840, 2160
def go(F, first, seg, rec="1"):
    if seg: os.environ["QOIMI_SEG_BYTES"] = str(seg)
    else: os.environ.pop("QOIMI_SEG_BYTES", None)
    os.environ["QOIMI_DEC_REC"] = rec
    cr = api.Context(0)
    dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    npx = w * h; desc = api.QoiDesc(w, h, 4, 0)
    ps = (npx * 4 + 255) // 256 * 256; ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
    px = torch.empty(F * ps, dtype=torch.uint8, device=dev); stt = torch.empty(F * ss, dtype=torch.uint8, device=dev)
    d1 = torch.full((F * ps,), 0xAB, dtype=torch.uint8, device=dev)
    lens = torch.zeros(F, dtype=torch.int32, device=dev)
    cr.synth_frames(synth.KIND_ID["uiflat"], synth.DEFAULT_SEED, first, F, w, h, px.data_ptr(), ps, st)
    cr.encode_batch(px.data_ptr(), ps, desc, F, stt.data_ptr(), ss, lens.data_ptr(), st); cr.encode_status(st)
    sizes = [int(x) for x in lens.cpu().numpy()]
    cr.decode_batch(stt.data_ptr(), ss, sizes, [desc] * F, 4, d1.data_ptr(), ps, st); torch.cuda.synchronize()
    a = d1.view(F, ps)[:, :npx * 4]; o = px.view(F, ps)[:, :npx * 4]
    ok = bool(torch.equal(a, o))
    print(f"F={F} first={first} seg={seg} rec={rec}", cr.decode_stats(), "exact", ok, flush=True)
    if not ok:
        ai = a.contiguous().view(-1).view(torch.int32).view(F, npx); oi = o.contiguous().view(-1).view(torch.int32).view(F, npx)
        badf = (ai != oi).any(dim=1).nonzero().flatten().tolist()
        print("  bad frames", badf[:20], "of", len(badf))
        for f in badf[:3]:
            bad = (ai[f] != oi[f]).nonzero().flatten()
            i0 = int(bad[0]); vals = torch.unique(ai[f][bad]).tolist()[:6]
            print(f"  frame {f}: {bad.numel()} bad px, first {i0} last {int(bad[-1])} got values {[hex(v & 0xFFFFFFFF) for v in vals]} size {sizes[f]}")
for a in [(256, 0, None), (256, 0, 512), (256, 0, 2048), (1, 45, 512), (4, 44, 512), (64, 0, 512), (256, 0, 512, "0")]:
    go(*a)
