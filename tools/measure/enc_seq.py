"""Diagnostic: batches of different content one after the other on ONE context, every encode call timed on its own (wall clock, synchronised)
and with its kernel times - what bench.py's class loop does, call by call.  usage: python tools/measure/enc_seq.py [frames] kind kind ..."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from qoi_amd import api, synth
F = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
kinds = sys.argv[2:] or ["noise", "constant", "uiflat", "photo_hard", "sprite_alpha"]
w, h = 3840, 2160
c = api.Context(0)
npx = w * h
ps = npx * 4
ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(F * ps, dtype=torch.uint8, device='cuda'); st = torch.empty(F * ss, dtype=torch.uint8, device='cuda'); lens = torch.zeros(F, dtype=torch.int32, device='cuda')
s = torch.cuda.current_stream().cuda_stream
desc = api.QoiDesc(w, h, 4, 0)
for kind in kinds:
    c.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 0, F, w, h, px.data_ptr(), ps, s)
    torch.cuda.synchronize()
    if os.environ.get("ASYNC") == "1":                     # bench.py's form: one call + status, then three calls back to back, one wait behind them
        c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s); c.encode_status(s)
        for rep in range(3):
            c.set_profiling(True)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s)
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
            prof = c.get_profile(s); c.set_profiling(False)
            print(kind, "three calls back to back, per call: wall ms", round(wall, 3), {k: round(v[0] / 3, 3) for k, v in prof.items() if v[1] and v[0] > 0.02}, flush=True)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(3):
                c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s)
            torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 3 * 1e3
            print(kind, "three calls back to back, no profiling, per call: wall ms", round(wall, 3), flush=True)
        continue
    for call in range(5):
        c.set_profiling(True)
        t0 = time.perf_counter()
        c.encode_batch(px.data_ptr(), ps, desc, F, st.data_ptr(), ss, lens.data_ptr(), s)
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) * 1e3
        prof = c.get_profile(s)
        c.set_profiling(False)
        print(kind, "call", call, "wall ms", round(wall, 3), {k: round(v[0], 3) for k, v in prof.items() if v[1] and v[0] > 0.02}, "workspace GB", round(c.workspace_bytes()["encode"] / 1e9, 3), flush=True)
