#!/usr/bin/env python
"""Diagnostic: does decoding two sub-batches CONCURRENTLY (two contexts, two streams, phase-shifted) beat decoding them one after
the other?  The transcoder is bound by vector instructions, P3 / P4 by LDS capacity: on paper they overlap."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from qoi_amd import api, synth  # noqa: E402

F = int(sys.argv[1]) if len(sys.argv) > 1 else 512          # frames in all
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
w, h = 3840, 2160
npx = w * h
ctx = [api.Context(0), api.Context(0)]
desc = api.QoiDesc(w, h, 4, api.QOI_SRGB)
ps = (npx * 4 + 255) // 256 * 256
ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(F * ps, dtype=torch.uint8, device="cuda")
st = torch.empty(F * ss, dtype=torch.uint8, device="cuda")
out = torch.empty(F * ps, dtype=torch.uint8, device="cuda")
lens = torch.zeros(F, dtype=torch.int32, device="cuda")
s0 = torch.cuda.current_stream().cuda_stream
ctx[0].synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, s0)
ctx[0].encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s0)
ctx[0].encode_status(s0)
sizes = [int(x) for x in lens.cpu().numpy()]
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
per = F // parts


def dec(c, k, stream):
    lo = k * per
    ctx[c].decode_batch(st.data_ptr() + lo * ss, ss, sizes[lo:lo + per], [desc] * per, 4, out.data_ptr() + lo * ps, ps, stream)


def sequential():
    for k in range(parts):
        dec(0, k, streams[0].cuda_stream)


def concurrent(delay):
    def worker(c):
        if c:
            time.sleep(delay)
        for k in range(c, parts, 2):
            dec(c, k, streams[c].cuda_stream)
    ts = [threading.Thread(target=worker, args=(c,)) for c in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


whole = timed(lambda: ctx[0].decode_batch(st.data_ptr(), ss, sizes, [desc] * F, 4, out.data_ptr(), ps, streams[0].cuda_stream))
print(f"{F} frames in one call: {whole:.2f} ms")
print(f"{parts} parts one after the other: {timed(sequential):.2f} ms")
for delay in (0.0, 0.002, 0.004):
    print(f"{parts} parts on two streams (second starts {delay * 1e3:.0f} ms later): {timed(lambda: concurrent(delay)):.2f} ms")
print("equal", bool(torch.equal(out.view(F, ps)[:, :npx * 4], px.view(F, ps)[:, :npx * 4])))
