"""GPU diagnostic: look-back placement on flat UI frames, fresh context per mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qoi_amd import api, synth
from gpu_util import DeviceBatch
from oracle import oracle_py
ref = oracle_py.load_ref() or oracle_py.load_port()
w, h, n = 1024, 600, 3
frames = [synth.frame_rgba("uiflat", w, h, 58 + i) for i in range(n)]
want = [ref.encode(f, w, h, 4) for f in frames]
for env in ({"QOIMI_ENC_LOOKBACK": "0"}, {"QOIMI_ENC_LOOKBACK": "1"}, {"QOIMI_ENC_LOOKBACK": "1", "QOIMI_ENC_WARM": "0"}, {"QOIMI_ENC_LOOKBACK": "1", "QOIMI_ENC_TICKET": "0"}):
    for k in ("QOIMI_ENC_LOOKBACK", "QOIMI_ENC_WARM", "QOIMI_ENC_TICKET"): os.environ.pop(k, None)
    os.environ.update(env)
    c = api.Context(0)
    b = DeviceBatch(c, w, h, 4, n)
    for i in range(n): b.upload(i, frames[i])
    for rep in range(3):
        lens = b.encode(); torch.cuda.synchronize()
        res = []
        for i in range(n):
            got = b.stream_bytes(i, lens[i])
            res.append("ok" if got == want[i] else f"BAD(len {len(got)} vs {len(want[i])})")
        print(env, "rep", rep, res, flush=True)
    c.close()
