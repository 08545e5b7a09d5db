#!/usr/bin/env python
"""Differential fuzz harness for the GPU decoder.

Input convention of the reference's fuzz target (qoifuzz.c:20-32): the first 4 bytes of a test input are the
`channels` argument (little-endian int), the rest is the stream handed to qoi_decode().  Where qoifuzz only
checks memory safety, this harness checks RESULTS: the drop-in qoi_decode of libqoi_mi355x.so must agree with
the oracle (the unmodified reference where oracle/_ref is built, else the C restatement) on NULL-ness, on the
desc it fills and on every pixel.

    python tests/fuzz_decode.py --iters 2000 --seed 1            # needs an MI355X
    python tests/fuzz_decode.py --replay crash.bin               # one input in qoifuzz's format

Mutations start from encoder-made streams (all content classes, both channel counts) and apply byte flips,
chunk-soup splices, truncations, header edits and size lies; pixel counts are capped so that a mutated header
cannot ask for gigabytes.
"""
from __future__ import annotations

import argparse
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MAX_PIXELS = 1 << 20


def seeds():
    from oracle import oracle_py
    from qoi_amd import synth
    port = oracle_py.load_port()
    out = []
    for kind in synth.KINDS:
        for (w, h) in ((61, 37), (200, 90), (512, 64)):
            out.append(port.encode(synth.frame_rgba(kind, w, h, 7), w, h, 4))
            out.append(port.encode(synth.frame_rgb(kind, w, h, 8), w, h, 3))
    return out


def mutate(rng: np.random.Generator, base: bytes) -> bytes:
    b = bytearray(base)
    for _ in range(int(rng.integers(1, 6))):
        op = int(rng.integers(0, 7))
        if op == 0 and len(b) > 22:                                   # byte flips in the chunk region
            for _ in range(int(rng.integers(1, 8))):
                b[int(rng.integers(14, len(b)))] = int(rng.integers(0, 256))
        elif op == 1 and len(b) > 30:                                 # truncate
            del b[int(rng.integers(22, len(b))):]
        elif op == 2 and len(b) > 22:                                 # splice random chunk soup
            at = int(rng.integers(14, len(b)))
            b[at:at] = bytes(rng.integers(0, 256, size=int(rng.integers(1, 64)), dtype=np.uint8))
        elif op == 3:                                                 # header edit (kept small)
            w = int(rng.integers(1, 700)); h = int(rng.integers(1, 700))
            b[4:12] = struct.pack(">II", w, h)
        elif op == 4:                                                 # channels / colorspace bytes
            b[12] = int(rng.choice([3, 4, 4, 3, 0, 5])); b[13] = int(rng.choice([0, 1, 0, 2]))
        elif op == 5 and len(b) > 40:                                 # duplicate a slice (long runs of ops)
            lo = int(rng.integers(14, len(b) - 8)); hi = min(len(b), lo + int(rng.integers(1, 200)))
            b[lo:lo] = b[lo:hi] * int(rng.integers(1, 4))
        elif op == 6:                                                 # magic
            b[int(rng.integers(0, 4))] ^= 1 << int(rng.integers(0, 8))
    return bytes(b)


def check(api, oracle, channels: int, stream: bytes):
    """Returns None if GPU and oracle agree, else a description."""
    if len(stream) >= 12:
        w, h = struct.unpack(">II", stream[4:12])
        if w * h > MAX_PIXELS:
            return None                                               # out of the harness' size budget
    want_px, want_desc = oracle.decode(stream, channels)
    got_px, got_desc = api.qoi_decode(stream, channels)
    if (want_px is None) != (got_px is None):
        return f"NULL-ness differs: oracle {'NULL' if want_px is None else 'ok'}, gpu {'NULL' if got_px is None else 'ok'}"
    if len(stream) >= 22 and channels in (0, 3, 4):
        a = (got_desc.width, got_desc.height, got_desc.channels, got_desc.colorspace)
        wd = (want_desc.width, want_desc.height, want_desc.channels, want_desc.colorspace)
        if a != wd:
            return f"desc differs: oracle {wd}, gpu {a}"
    if want_px is not None and not np.array_equal(want_px, got_px):
        return f"pixels differ at byte {int(np.argmax(want_px != got_px))}"
    return None


def run(iters: int, seed: int, api=None, oracle=None, out_dir: str | None = None) -> int:
    import torch  # noqa: F401  (first, so the library binds to torch's HIP runtime)
    from oracle import oracle_py
    from qoi_amd import api as _api
    api = api or _api
    oracle = oracle or oracle_py.load_ref() or oracle_py.load_port()
    rng = np.random.default_rng(seed)
    base = seeds()
    bad = 0
    for it in range(iters):
        stream = mutate(rng, base[int(rng.integers(0, len(base)))])
        channels = int(rng.choice([0, 3, 4, 4, 0, 1, 5]))
        msg = check(api, oracle, channels, stream)
        if msg:
            bad += 1
            name = f"fuzz_fail_{seed}_{it}.bin"
            if out_dir:
                with open(os.path.join(out_dir, name), "wb") as f:
                    f.write(struct.pack("<i", channels) + stream)      # qoifuzz.c input convention
            print(f"[{it}] channels={channels} size={len(stream)}: {msg} ({name})", flush=True)
    return bad


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=1000)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--replay", help="one input in qoifuzz's format (int channels + stream)")
    ap.add_argument("--out", default=".")
    a = ap.parse_args()
    if a.replay:
        import torch  # noqa: F401
        from oracle import oracle_py
        from qoi_amd import api
        data = open(a.replay, "rb").read()
        msg = check(api, oracle_py.load_ref() or oracle_py.load_port(), struct.unpack("<i", data[:4])[0], data[4:])
        print(msg or "agree")
        sys.exit(1 if msg else 0)
    bad = run(a.iters, a.seed, out_dir=a.out)
    print(f"{a.iters} inputs, {bad} disagreements")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
