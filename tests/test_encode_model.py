"""The slab-parallel closed form (tools/encode_model.py) reproduces the reference bytes."""
import numpy as np

import cases
from tools import encode_model as M


def test_model_matches_golden(golden):
    for c in cases.encode_cases():
        if c["w"] * c["h"] > 4000:
            continue
        px = c["pixels"].reshape(-1, c["ch"])
        if c["ch"] == 3:
            px = np.concatenate([px, np.full((len(px), 1), 255, np.uint8)], axis=1)
        u = np.ascontiguousarray(px).view(np.uint32).reshape(-1)
        want = golden[f"enc/{c['name']}/stream"].tobytes()[14:-8]
        for slab in (1, 7, 64, 1000):
            assert M.encode_chunks(u, slab) == want, (c["name"], slab)
