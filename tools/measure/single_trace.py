#!/usr/bin/env python
"""Diagnostic: N single-frame encode + decode calls (device-resident 4K photo frame) for a rocprofv3 --kernel-trace run.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/measure/single_trace.py [reps]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libsel  # noqa: E402
libsel.use_env_library()            # QOIMI_TOOLS_LIB=<path>: the build under test
import torch  # noqa: E402
from qoi_amd import api, synth  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
mode = sys.argv[2] if len(sys.argv) > 2 else "both"
w, h = int(os.environ.get("W", 3840)), int(os.environ.get("H", 2160))
npx = w * h
ctx = api.Context(0)
stream = torch.cuda.current_stream().cuda_stream
desc = api.QoiDesc(w, h, 4, api.QOI_SRGB)
ps = (npx * 4 + 255) // 256 * 256
ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
px = torch.empty(ps, dtype=torch.uint8, device="cuda")
st = torch.empty(ss, dtype=torch.uint8, device="cuda")
out = torch.empty(ps, dtype=torch.uint8, device="cuda")
lens = torch.zeros(16, dtype=torch.int32, device="cuda")
ctx.synth_frames(synth.KIND_ID[os.environ.get("KIND", "photo")], synth.DEFAULT_SEED, 0, 1, w, h, px.data_ptr(), ps, stream)
ctx.encode_batch(px.data_ptr(), ps, desc, 1, st.data_ptr(), ss, lens.data_ptr(), stream)
ctx.encode_status(stream)
n = [int(lens[0].item())]
for _ in range(3):
    ctx.encode_batch(px.data_ptr(), ps, desc, 1, st.data_ptr(), ss, lens.data_ptr(), stream)
    ctx.decode_batch(st.data_ptr(), ss, n, [desc], 4, out.data_ptr(), ps, stream)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps if mode != "dec" else 0):
    ctx.encode_batch(px.data_ptr(), ps, desc, 1, st.data_ptr(), ss, lens.data_ptr(), stream)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(reps if mode != "enc" else 0):
    ctx.decode_batch(st.data_ptr(), ss, n, [desc], 4, out.data_ptr(), ps, stream)
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"single frame: encode {(t1 - t0) / reps * 1e6:.1f} us, decode {(t2 - t1) / reps * 1e6:.1f} us, equal {bool(torch.equal(out[:npx * 4], px[:npx * 4]))}")
if os.environ.get("STATS"):
    print("decode stats of the last call:", ctx.decode_stats())
    ctx.set_profiling(True)
    ctx.decode_batch(st.data_ptr(), ss, n, [desc], 4, out.data_ptr(), ps, stream)
    prof = ctx.get_profile(stream)
    print("per kernel (us, launches):", {k: (round(v[0] * 1e3, 1), v[1]) for k, v in prof.items() if v[1]})
