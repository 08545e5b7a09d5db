"""GPU diagnostic: the bench's other_content sequence (shared buffers, F frames) on the record pipeline; reports mismatches."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from qoi_amd import api, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 256
kinds = sys.argv[2].split(",") if len(sys.argv) > 2 else ["photo", "noise", "constant", "uiflat"]
w, h = 3840, 2160
cr = api.Context(0)
dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
npx = w * h; desc = api.QoiDesc(w, h, 4, 0)
ps = (npx * 4 + 255) // 256 * 256; ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(F * ps, dtype=torch.uint8, device=dev); stt = torch.empty(F * ss, dtype=torch.uint8, device=dev)
d1 = torch.zeros(F * ps, dtype=torch.uint8, device=dev)
lens = torch.zeros(F, dtype=torch.int32, device=dev)
for kind in kinds:
    cr.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, st)
    cr.encode_batch(px.data_ptr(), ps, desc, F, stt.data_ptr(), ss, lens.data_ptr(), st); cr.encode_status(st)
    sizes = [int(x) for x in lens.cpu().numpy()]
    for rep in range(2):
        cr.decode_batch(stt.data_ptr(), ss, sizes, [desc] * F, 4, d1.data_ptr(), ps, st); torch.cuda.synchronize()
        a = d1.view(F, ps)[:, :npx * 4]; o = px.view(F, ps)[:, :npx * 4]
        ok = bool(torch.equal(a, o))
        print(kind, "rep", rep, cr.decode_stats(), "exact", ok, flush=True)
        if not ok:
            ai = a.contiguous().view(-1).view(torch.int32).view(F, npx); oi = o.contiguous().view(-1).view(torch.int32).view(F, npx)
            badf = (ai != oi).any(dim=1).nonzero().flatten().tolist()
            print("  bad frames", badf[:20], "of", len(badf))
            f = badf[0]; bad = (ai[f] != oi[f]).nonzero().flatten()
            i0 = int(bad[0]); print(f"  frame {f}: {bad.numel()} bad px, first {i0} (row {i0 // w} col {i0 % w}) last {int(bad[-1])} got {int(ai[f, i0]) & 0xFFFFFFFF:08x} want {int(oi[f, i0]) & 0xFFFFFFFF:08x}")
            d = bad[1:] - bad[:-1]
            print("    bad runs start at", torch.cat([bad[:1], bad[1:][d > 1]])[:8].tolist(), "len sizes", sizes[f])
