"""GPU diagnostic: encoder parity sweep over set sizes, placements and MIXED content (stretches of noise inside photographs:
sets that spill part of their bytes), against the reference encoder.  usage: python tools/dev/sweep_enc.py [seed]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qoi_amd import api, synth
from gpu_util import DeviceBatch
from oracle import oracle_py
ref = oracle_py.load_ref() or oracle_py.load_port()
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 11
bad = 0; n_checked = 0; t0 = time.time()
shapes = [(2560, 1440), (1920, 1080), (1000, 999), (4096, 33), (640, 481)]


def mixed(rng, w, h, ch, s):
    a = synth.frame_rgba("photo", w, h, s).reshape(-1, 4).copy()
    b = synth.frame_rgba("noise", w, h, s + 1).reshape(-1, 4)
    n = a.shape[0]; pos = 0
    while pos < n:                                     # alternate stretches: photo, noise, flat, alpha steps
        L = int(rng.integers(1, 6000)); k = int(rng.integers(0, 5))
        if k == 1: a[pos:pos + L] = b[pos:pos + L]
        elif k == 2: a[pos:pos + L] = a[pos]
        elif k == 3: a[pos:pos + L, 3] = rng.integers(0, 256)
        elif k == 4: a[pos:pos + L:2] = b[pos:pos + L:2]
        pos += L
    return np.ascontiguousarray(a.reshape(h, w, 4)[:, :, :ch])


for env in ({"QOIMI_ENC_SET_SLABS": "1"}, {"QOIMI_ENC_SET_SLABS": "2"}, {"QOIMI_ENC_SET_SLABS": "3"}, {"QOIMI_ENC_SET_SLABS": "4"}, {"QOIMI_ENC_SET_SLABS": "8"},
            {"QOIMI_ENC_SET_SLABS": "3", "QOIMI_ENC_LOOKBACK": "0"}, {"QOIMI_ENC_SET_SLABS": "5", "QOIMI_ENC_LOOKBACK": "1"}, {}):
    for k in ("QOIMI_ENC_SET_SLABS", "QOIMI_ENC_LOOKBACK"):
        os.environ.pop(k, None)
    os.environ.update(env)
    c = api.Context(0)
    rng = np.random.default_rng(seed)
    for (w, h) in shapes:
        for ch in (4, 3):
            n = 10
            frames = [mixed(rng, w, h, ch, 1000 + i) for i in range(n)]
            b = DeviceBatch(c, w, h, ch, n)
            for i in range(n): b.upload(i, frames[i])
            lens = b.encode(); torch.cuda.synchronize()
            for i in range(n):
                want = ref.encode(frames[i], w, h, ch)
                ok = b.stream_bytes(i, lens[i]) == want
                n_checked += 1
                if not ok:
                    bad += 1; print("MISMATCH", env, w, h, ch, i, int(lens[i]), len(want), flush=True)
    c.close()
print("checked", n_checked, "bad", bad, "sec", round(time.time() - t0, 1))
