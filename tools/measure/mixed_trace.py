#!/usr/bin/env python
"""Diagnostic: bench.py's mixed_directory leg on its own (288 images, 64 shapes, six content classes) for a rocprofv3 --kernel-trace run.
usage (GPU box): rocprofv3 --kernel-trace --output-format csv -d OUT -o t -- python tools/measure/mixed_trace.py"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import libsel  # noqa: E402
libsel.use_env_library()
import torch  # noqa: E402
import bench  # noqa: E402
from qoi_amd import api, synth  # noqa: E402

ctx = api.Context(0)
stream = torch.cuda.current_stream().cuda_stream
pixels = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")
streams = torch.empty(5 << 30, dtype=torch.uint8, device="cuda")
decoded = torch.empty(4 << 30, dtype=torch.uint8, device="cuda")
lens = torch.zeros(4096, dtype=torch.int32, device="cuda")


def timed(fn, reps):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps


r = bench.mixed_directory_leg(torch, api, synth, ctx, pixels, streams, decoded, lens, stream, timed)
print(json.dumps({k: r[k] for k in r if k not in ("workload", "note", "hash_check")}))
print("decode stats:", ctx.decode_stats())
