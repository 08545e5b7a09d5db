"""qoi_encode / qoi_decode on host pointers, one 4K photo frame (bench.py's dropin_host_pointers alone): python tools/measure/dropin_time.py
- run once per setting of QOIMI_DROPIN_SPLIT etc.; the environment is read when the library loads."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
from qoi_amd import api, synth
w, h = 3840, 2160
c = api.Context(0)
px = torch.empty(w * h * 4, dtype=torch.uint8, device='cuda')
c.synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 0, 1, w, h, px.data_ptr(), w * h * 4, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
best = None
for _ in range(3):
    r = bench.dropin_path(torch, api, c, px, w, h, 'cuda')
    if best is None or r["decode_ms"] < best["decode_ms"]:
        best = r
print(os.environ.get("QOIMI_DROPIN_SPLIT", "-"), json.dumps({k: best[k] for k in ("encode_ms", "decode_ms", "copies_alone_ms", "frac_of_copies", "round_trip_exact")}))
