#!/usr/bin/env python
"""Seed corpus for tests/fuzz/qoi_fuzz_diff: every stream of the golden vectors (tests/golden/qoi_golden.npz - reference-made
streams, the Appendix-B edge cases, the seeded mutations) in qoifuzz.c's input convention: a 4-byte `channels` argument
(little-endian int) in front of the stream.  usage: make_corpus.py OUT_DIR"""
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main(out):
    os.makedirs(out, exist_ok=True)
    g = np.load(os.path.join(ROOT, "tests", "golden", "qoi_golden.npz"))
    n = 0
    for k in g.files:
        if not k.endswith("/stream"):
            continue
        s = g[k].tobytes()
        if len(s) > 16384:
            continue
        for ch in ((0, 3, 4) if k.startswith("enc/") else (4,)):
            with open(os.path.join(out, f"seed_{n:04d}"), "wb") as f:
                f.write(struct.pack("<i", ch) + s)
            n += 1
    # streams dense in QOI_OP_RGB / QOI_OP_RGBA (the decoder's record pairs), short enough for the harness's -max_len
    rng = np.random.default_rng(4)
    for p_rgba in (1.0, 0.5, 0.0):
        for p_hi in (1.0, 0.9):
            body = bytearray()
            npx = 0
            while len(body) < 6000:
                if rng.random() < p_hi:
                    v = rng.integers(0, 256, size=4)
                    body += bytes([0xFF, v[0], v[1], v[2], v[3]]) if rng.random() < p_rgba else bytes([0xFE, v[0], v[1], v[2]])
                else:
                    body.append(int(rng.integers(0, 0xFE)) & 0x7F)       # INDEX or DIFF
                npx += 1
            w = 61
            s = b"qoif" + struct.pack(">II", w, (npx + w - 1) // w) + bytes([4, 0]) + bytes(body) + bytes(7) + b"\x01"
            for ch in (3, 4):
                with open(os.path.join(out, f"seed_{n:04d}"), "wb") as f:
                    f.write(struct.pack("<i", ch) + s)
                n += 1
    print(n, "seed inputs in", out)


if __name__ == "__main__":
    main(sys.argv[1])
