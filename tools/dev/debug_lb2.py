"""GPU diagnostic: one flat UI frame that the encoder gets wrong - alone, in different batch positions, with the order-free probe."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from qoi_amd import api, synth
from gpu_util import DeviceBatch
from oracle import oracle_py
ref = oracle_py.load_ref() or oracle_py.load_port()
w, h = 1024, 600
bad = synth.frame_rgba("uiflat", w, h, 60)
other = synth.frame_rgba("uiflat", w, h, 58)
want_bad, want_other = ref.encode(bad, w, h, 4), ref.encode(other, w, h, 4)
def run(frames, wants, env):
    for k in ("QOIMI_ENC_LOOKBACK", "QOIMI_ENC_WARM", "QOIMI_ENC_PROBE"): os.environ.pop(k, None)
    os.environ.update(env)
    c = api.Context(0); n = len(frames)
    b = DeviceBatch(c, w, h, 4, n)
    for i in range(n): b.upload(i, frames[i])
    lens = b.encode(); torch.cuda.synchronize()
    res = []
    for i in range(n):
        got = b.stream_bytes(i, lens[i])
        if got == wants[i]: res.append("ok")
        else:
            k = next(j for j in range(min(len(got), len(wants[i]))) if got[j] != wants[i][j])
            res.append(f"BAD@{k}")
    print(env, [("bad" if f is bad else "other") for f in frames], res, flush=True)
    c.close()
base = {"QOIMI_ENC_LOOKBACK": "0"}
run([bad], [want_bad], base)
run([bad, other], [want_bad, want_other], base)
run([other, bad], [want_other, want_bad], base)
run([other, other, bad], [want_other, want_other, want_bad], base)
run([bad], [want_bad], {**base, "QOIMI_ENC_PROBE": "0"})
run([bad], [want_bad], {**base, "QOIMI_ENC_WARM": "0"})
run([bad], [want_bad], {**base, "QOIMI_ENC_WARM": "0", "QOIMI_ENC_PROBE": "0"})
