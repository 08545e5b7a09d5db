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
    print(n, "seed inputs in", out)


if __name__ == "__main__":
    main(sys.argv[1])
