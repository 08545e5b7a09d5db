"""GPU diagnostic: a broader seeded parity sweep than the test suite's (encode bytes and decode pixels against the reference)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qoi_amd import api, synth
from gpu_util import DeviceBatch
from oracle import oracle_py
ref = oracle_py.load_ref() or oracle_py.load_port()
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
c = api.Context(0)
bad = 0; n_checked = 0; t0 = time.time()
shapes = [(2560, 1440), (1920, 1080), (3840, 400), (777, 1333), (4096, 64), (1280, 720), (333, 333), (96, 64), (97, 65)]
for (w, h) in shapes:
    for ch in (4, 3):
        n = 4
        kinds = ["uiflat", "uiflat", synth.KINDS[int(rng.integers(0, 4))], "constant" if rng.integers(0, 2) else "photo"]
        seeds = [int(x) for x in rng.integers(0, 1 << 20, n)]
        frames = [np.ascontiguousarray(synth.frame_rgba(kinds[i], w, h, seeds[i])[:, :, :ch]) for i in range(n)]
        b = DeviceBatch(c, w, h, ch, n)
        for i in range(n): b.upload(i, frames[i])
        lens = b.encode(); torch.cuda.synchronize()
        out = torch.full((n * b.pixel_stride,), 0xCD, dtype=torch.uint8, device="cuda")
        stride = b.decode_into(out, lens, ch); got = out.cpu().numpy()
        for i in range(n):
            want = ref.encode(frames[i], w, h, ch)
            e_ok = b.stream_bytes(i, lens[i]) == want
            d_ok = np.array_equal(got[i * stride:i * stride + w * h * ch], frames[i].reshape(-1))
            n_checked += 1
            if not (e_ok and d_ok):
                bad += 1; print("MISMATCH", w, h, ch, kinds[i], seeds[i], "encode" if not e_ok else "", "decode" if not d_ok else "", flush=True)
print("checked", n_checked, "bad", bad, "rounds", c.decode_stats(), "sec", round(time.time() - t0, 1))
