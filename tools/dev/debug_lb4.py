import os, sys, struct
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from qoi_amd import api, synth
from gpu_util import DeviceBatch
import encode_model as em
w, h = 1024, 600
bad = synth.frame_rgba("uiflat", w, h, 60)
os.environ["QOIMI_ENC_LOOKBACK"] = "0"; os.environ["QOIMI_ENC_WARM"] = "0"
os.environ["QOIMI_ENC_DEBUG_DUMP"] = "/tmp/encdump.bin"
c = api.Context(0)
b = DeviceBatch(c, w, h, 4, 1)
b.upload(0, bad); b.encode(); torch.cuda.synchronize()
raw = open("/tmp/encdump.bin", "rb").read()
T, G, spi, gpi = struct.unpack("<4Q", raw[:32]); o = 32
def take(n, dt):
    global o
    a = np.frombuffer(raw, dtype=dt, count=n, offset=o); o += a.nbytes; return a
sum_tab = take(T * 64, np.uint32).reshape(T, 64); sum_valid = take(T, np.uint64)
ent_tab = take(T * 64, np.uint32).reshape(T, 64); ent_valid = take(T, np.uint64)
px = bad.reshape(-1, 4).copy().view(np.uint32).reshape(-1)
tab, valid, le = em.slab_summaries(px, 1024)
etab, ele = em.scan_entries(tab, valid, le)
for s in (320, 321, 322):
    k = np.nonzero(sum_tab[s] != etab[s])[0]
    print("slab", s, "start table differs at slots", k, [hex(x) for x in sum_tab[s][k]], "want", [hex(x) for x in etab[s][k]])
    print("   step0 edges", hex(int(ent_valid[s])), "seen[lane0]", hex(ent_tab[s][0]), "px", hex(px[s * 1024]), "table[62] at start", hex(sum_tab[s][62]))
