"""GPU diagnostic: entry-state arrays of the encoder's generic path against numpy, for the flat UI frame it gets wrong."""
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
grp_tab = take(G * 64, np.uint32).reshape(G, 64); grp_valid = take(G, np.uint64)
gent_tab = take(G * 64, np.uint32).reshape(G, 64)
px = bad.reshape(-1, 4).copy().view(np.uint32).reshape(-1)
tab, valid, le = em.slab_summaries(px, 1024)
bits = lambda m: np.array([(int(m) >> k) & 1 for k in range(64)], dtype=bool)
bad_s = [s for s in range(T) if not (np.array_equal(bits(sum_valid[s]), valid[s]) and np.array_equal(sum_tab[s][valid[s]], tab[s][valid[s]]))]
print("T", T, "G", G, "slab summaries differing:", bad_s[:10], len(bad_s))
for s in bad_s[:3]:
    dv = bits(sum_valid[s]); k = np.nonzero((dv != valid[s]) | ((sum_tab[s] != tab[s]) & valid[s]))[0]
    print("  slab", s, "slots", k[:8], "dev valid", dv[k][:8], "want valid", valid[s][k][:8], "dev", [hex(x) for x in sum_tab[s][k][:8]], "want", [hex(x) for x in tab[s][k][:8]])
# expected full entry table per slab
etab, ele = em.scan_entries(tab, valid, le)
for s in (320, 321, 322):
    g = s // 64
    dev = np.where(bits(ent_valid[s]), ent_tab[s], gent_tab[g])
    k = np.nonzero(dev != etab[s])[0]
    print("slab", s, "entry slots differing", k, [hex(x) for x in dev[k]], [hex(x) for x in etab[s][k]], "loc valid", bits(ent_valid[s])[k], "gent", [hex(x) for x in gent_tab[g][k]], "grp_valid(g-1)", bits(grp_valid[g - 1])[k], "grp_tab(g-1)", [hex(x) for x in grp_tab[g - 1][k]])
