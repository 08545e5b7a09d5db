"""GPU diagnostic: encode / decode time of a batch of 3-channel frames beside the same frames with 4 channels.
usage: python tools/dev/rgb_time.py [frames]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qoi_amd import api, synth
from gpu_util import DeviceBatch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
w, h = 3840, 2160
c = api.Context(0)
b4 = DeviceBatch(c, w, h, 4, n)
c.synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 0, n, w, h, b4.pixels.data_ptr(), b4.pixel_stride, b4.stream)
torch.cuda.synchronize()
b3 = DeviceBatch(c, w, h, 3, n)
p4 = b4.pixels.view(n, -1)[:, :w * h * 4].reshape(n, w * h, 4)
b3.pixels.view(n, -1)[:, :w * h * 3].copy_(p4[:, :, :3].reshape(n, -1))
torch.cuda.synchronize()
for name, b in (("rgba", b4), ("rgb ", b3), ("rgba", b4), ("rgb ", b3)):
    lens = b.encode(); torch.cuda.synchronize()
    out = torch.empty(n * b.pixel_stride, dtype=torch.uint8, device="cuda")
    b.decode_into(out, lens); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        c.encode_batch(b.pixels.data_ptr(), b.pixel_stride, b.desc, n, b.streams.data_ptr(), b.stream_stride, b.lens.data_ptr(), b.stream)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    for _ in range(5):
        b.decode_into(out, lens)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ok = torch.equal(out.view(n, -1)[:, :w * h * b.ch], b.pixels.view(n, -1)[:, :w * h * b.ch])
    print(name, "frames", n, "encode ms", round((t1 - t0) / 5 * 1e3, 3), "decode ms", round((t2 - t1) / 5 * 1e3, 3), "bytes/px", round(float(lens.sum()) / (n * w * h), 4), "round trip", ok)
