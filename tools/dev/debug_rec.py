"""GPU diagnostic: decode the same streams with the record pipeline and with the round-1 byte-stream passes, report the first differences."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from qoi_amd import api, synth

def run(kind, F, w, h, seg=None):
    if seg: os.environ["QOIMI_SEG_BYTES"] = str(seg)
    else: os.environ.pop("QOIMI_SEG_BYTES", None)
    os.environ["QOIMI_DEC_REC"] = "1"; cr = api.Context(0)
    os.environ["QOIMI_DEC_REC"] = "0"; co = api.Context(0)
    dev = torch.device("cuda", 0); st = torch.cuda.current_stream().cuda_stream
    npx = w * h
    desc = api.QoiDesc(w, h, 4, 0)
    ps = (npx * 4 + 255) // 256 * 256; ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
    px = torch.empty(F * ps, dtype=torch.uint8, device=dev); stt = torch.empty(F * ss, dtype=torch.uint8, device=dev)
    d1 = torch.zeros(F * ps, dtype=torch.uint8, device=dev); d2 = torch.zeros(F * ps, dtype=torch.uint8, device=dev)
    lens = torch.zeros(F, dtype=torch.int32, device=dev)
    cr.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, st)
    cr.encode_batch(px.data_ptr(), ps, desc, F, stt.data_ptr(), ss, lens.data_ptr(), st); cr.encode_status(st)
    sizes = [int(x) for x in lens.cpu().numpy()]
    cr.decode_batch(stt.data_ptr(), ss, sizes, [desc] * F, 4, d1.data_ptr(), ps, st); torch.cuda.synchronize(); r1 = cr.decode_stats()
    co.decode_batch(stt.data_ptr(), ss, sizes, [desc] * F, 4, d2.data_ptr(), ps, st); torch.cuda.synchronize(); r2 = co.decode_stats()
    a = d1.view(F, -1)[:, :npx * 4].view(torch.int32).view(F, npx) if False else d1.view(F, ps)[:, :npx * 4].contiguous().view(-1).view(torch.int32).view(F, npx)
    b = d2.view(F, ps)[:, :npx * 4].contiguous().view(-1).view(torch.int32).view(F, npx)
    o = px.view(F, ps)[:, :npx * 4].contiguous().view(-1).view(torch.int32).view(F, npx)
    print(f"{kind} F={F} {w}x{h} seg={seg}: rec rounds {r1}, old rounds {r2}; rec==orig {bool(torch.equal(a, o))} old==orig {bool(torch.equal(b, o))}")
    for f in range(F):
        bad = (a[f] != o[f]).nonzero().flatten()
        if bad.numel():
            i0 = int(bad[0]); print(f"  frame {f}: {bad.numel()} bad px, first {i0} (row {i0 // w}, col {i0 % w}), last {int(bad[-1])}; got {int(a[f, i0]) & 0xFFFFFFFF:08x} want {int(o[f, i0]) & 0xFFFFFFFF:08x}")
            # runs of bad pixels
            d = bad[1:] - bad[:-1]
            starts = torch.cat([bad[:1], bad[1:][d > 1]])[:6].tolist()
            print("    bad runs start at", starts)
            s = bytes(stt[f * ss: f * ss + sizes[f]].cpu().numpy().tobytes())
            break

if __name__ == "__main__":
    for args in [("uiflat", 2, 3840, 2160, None), ("uiflat", 2, 3840, 2160, 2048), ("uiflat", 2, 3840, 2160, 128), ("uiflat", 32, 3840, 2160, 2048), ("uiflat", 4, 640, 480, 256)]:
        run(*args)
