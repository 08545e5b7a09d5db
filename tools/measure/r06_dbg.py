import os, sys, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["QOIMI_TUNING"] = "1"
import numpy as np, torch
import cases
from qoi_amd import api
name, stream, w, h = next(iter(cases.pair_streams()))
res = {}
for split in ("1", "0"):
    os.environ["QOIMI_DEC_SPLIT"] = split
    os.environ["QOIMI_DEC_DEBUG_DUMP"] = f"/tmp/dump{split}.bin"
    c = api.Context(0)
    s = torch.from_numpy(np.frombuffer(stream + b"\0" * 8, dtype=np.uint8).copy()).cuda()
    out = torch.full((w * h * 4 + 8,), 0xAB, dtype=torch.uint8, device="cuda")
    c.decode_batch(s.data_ptr(), s.numel(), [len(stream)], [api.QoiDesc(w, h, 4, 0)], 4, out.data_ptr(), w * h * 4)
    c.close()
    raw = open(f"/tmp/dump{split}.bin", "rb").read()
    total, sp, rows, B = struct.unpack("<4Q", raw[:32])
    o = 32
    gran = np.frombuffer(raw[o:o + total * 4], dtype=np.uint32); o += total * 4
    parse = np.frombuffer(raw[o:o + total * 24], dtype=np.uint32).reshape(total, 6); o += total * 24
    pxoff = np.frombuffer(raw[o:o + total * 4], dtype=np.uint32); o += total * 4
    fail = np.frombuffer(raw[o:o + total], dtype=np.uint8)
    res[split] = (gran, parse, pxoff, fail)
    print("split", split, "total", total, "tr_split", sp, "rows", rows, "fails", int(fail.sum()))
g1, p1, o1, f1 = res["1"]; g0, p0, o0, f0 = res["0"]
n1 = (g1 & 0xFFFF) + (g1 >> 16)
d = np.nonzero(p1[:, 1] != p0[:, 1])[0]
print("npix differs at segments", d[:10], "split", p1[d[:5], 1], "plain", p0[d[:5], 1])
d2 = np.nonzero(o1 != o0)[0]
print("px_off differs first at", d2[:5])
for q in (1076, 1077, 1078):
    print(q, "split gran", hex(int(g1[q])), "rows", int(n1[q]), "plain rows", int(g0[q]), "npix", int(p1[q, 1]), int(p0[q, 1]), "exit", hex(int(p1[q, 0])), hex(int(p0[q, 0])), "pxoff", int(o1[q]), int(o0[q]))
