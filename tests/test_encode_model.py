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


def test_model_on_images_opening_with_the_start_value(ref, port):
    """The closed form on what the GPU encoder got wrong until round 4 (tests/test_gpu_parity.py::test_start_value_runs_past_the_first_set):
    an image that opens with pixels of the start value {0,0,0,255} for longer than a slab, and meets that value again later - its first
    return is a literal chunk (the table slot is still zero), not QOI_OP_INDEX.  The model writes the table at edge pixels only, as
    qoi.h:430-436 does; the kernels' all-lanes probe had to learn the same."""
    oracle = ref or port
    rng = np.random.default_rng(5)
    for n0 in (5, 64, 65, 700, 1500):
        n = n0 + 600
        a = rng.integers(0, 256, size=(n, 4), dtype=np.uint8)
        a[:n0] = (0, 0, 0, 255)
        a[n0 + 1::53] = (0, 0, 0, 255)
        a[:, 3] = 255
        u = np.ascontiguousarray(a).view(np.uint32).reshape(-1)
        want = oracle.encode(a.reshape(1, n, 4), n, 1, 4)[14:-8]
        for slab in (1, 64, 256, 1000):
            assert M.encode_chunks(u, slab) == want, (n0, slab)
