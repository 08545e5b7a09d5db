#!/usr/bin/env python
"""Diagnostic: the mixed directory's shapes, one content class at a time: rounds and re-opened segments of qoimi_decode_batch."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from qoi_amd import api, synth

ctx = api.Context(0)
stream = torch.cuda.current_stream().cuda_stream
rng = np.random.default_rng(2026)
shapes = set()
while len(shapes) < 64:
    shapes.add((int(rng.integers(48, 2049)), int(rng.integers(48, 1537))))
shapes = sorted(shapes)
M = 48
for kind in ["photo", "noise", "uiflat", "constant", "photo_hard", "sprite_alpha"]:
    items = [shapes[(i * 7) % len(shapes)] for i in range(M)]
    po, off = [], 0
    for iw, ih in items:
        po.append(off); off += (iw * ih * 4 + 255) // 256 * 256
    ss = (max(api.encode_bound(iw, ih, 4) for iw, ih in items) + 255) // 256 * 256
    ps = (max(iw * ih * 4 for iw, ih in items) + 255) // 256 * 256
    pixels = torch.empty(off, dtype=torch.uint8, device="cuda")
    streams = torch.empty(M * ss, dtype=torch.uint8, device="cuda")
    decoded = torch.empty(M * ps, dtype=torch.uint8, device="cuda")
    lens = torch.zeros(M, dtype=torch.int32, device="cuda")
    descs = [api.QoiDesc(iw, ih, 4, 0) for iw, ih in items]
    for i, (iw, ih) in enumerate(items):
        ctx.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 40000 + i, 1, iw, ih, pixels.data_ptr() + po[i], iw * ih * 4, stream)
    ctx.encode_images(pixels.data_ptr(), po, descs, streams.data_ptr(), [i * ss for i in range(M)], lens.data_ptr(), stream)
    ctx.encode_status(stream)
    sizes = [int(x) for x in lens.cpu().numpy()]
    for _ in range(3):
        ctx.decode_batch(streams.data_ptr(), ss, sizes, descs, 4, decoded.data_ptr(), ps, stream)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ctx.decode_batch(streams.data_ptr(), ss, sizes, descs, 4, decoded.data_ptr(), ps, stream)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(kind, "decode ms", round(dt * 1e3, 3), "stream MB", round(sum(sizes) / 1e6, 1), ctx.decode_stats())
