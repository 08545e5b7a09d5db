#!/usr/bin/env python
"""Differential fuzz of the GPU DECODER on device batches (companion of tests/fuzz_decode.py, which drives the drop-in qoi_decode
with small mutated streams).  Here: streams of up to a few hundred kilobytes built from stretches of different character -
encoder-made bytes, uniform random bytes, and chunk soups biased towards one op (QOI_OP_INDEX on arbitrary slots, QOI_OP_RUN,
QOI_OP_RGBA, QOI_OP_LUMA ...) - i.e. valid-grammar streams no encoder would write, which defeat the decoder's slot / alpha
speculation and go through its verify-and-repair rounds; several per call, random segment size (QOIMI_SEG_BYTES), 3- and 4-channel
output.  Every image must equal the reference decoder's (qoi.h:488-590) pixel for pixel, including the pixels past a short stream
(qoi.h:544: the last pixel repeats) and the clipping of a long one.

    python tests/fuzz_decode_batch.py --iters 200 --seed 1            # needs an MI355X
"""
from __future__ import annotations

import argparse
import os
import struct
import sys
import time

import numpy as np

os.environ.setdefault("QOIMI_TUNING", "1")      # the placement / segment-size knobs below are looked at only under it (qoi_host.hip)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

END = bytes([0, 0, 0, 0, 0, 0, 0, 1])


def soup(rng: np.random.Generator, nbytes: int) -> bytes:
    """chunk bytes of one character"""
    style = int(rng.integers(0, 9))
    if style == 0:                                                    # anything
        return rng.integers(0, 256, size=nbytes, dtype=np.uint8).tobytes()
    if style == 1:                                                    # INDEX on arbitrary slots (most of them never written: zero pixels, or stale ones)
        return rng.integers(0, 64, size=nbytes, dtype=np.uint8).tobytes()
    if style == 2:                                                    # INDEX on few slots between runs
        b = rng.choice(np.array([3, 17, 53, 0, 63, 0xC0, 0xC5, 0xFD, 0xE0], dtype=np.uint8), size=nbytes)
        return b.tobytes()
    if style == 3:                                                    # runs of every length, 62-runs back to back
        return rng.integers(0xC0, 0xFE, size=nbytes, dtype=np.uint8).tobytes()
    if style == 4:                                                    # RGBA chunks: alpha moves all the time
        a = rng.integers(0, 256, size=(nbytes // 5 + 1, 5), dtype=np.uint8); a[:, 0] = 0xFF
        return a.tobytes()[:nbytes]
    if style == 5:                                                    # RGB chunks
        a = rng.integers(0, 256, size=(nbytes // 4 + 1, 4), dtype=np.uint8); a[:, 0] = 0xFE
        return a.tobytes()[:nbytes]
    if style == 6:                                                    # LUMA / DIFF: relative chunks only, wrap-around everywhere
        return rng.integers(0x40, 0xC0, size=nbytes, dtype=np.uint8).tobytes()
    if style == 7:                                                    # relative chunks with an INDEX now and then (slot speculation right until it is not)
        b = rng.integers(0x40, 0xC0, size=nbytes, dtype=np.uint8)
        k = rng.random(nbytes) < 0.02
        b[k] = rng.integers(0, 64, size=int(k.sum()), dtype=np.uint8)
        return b.tobytes()
    # RGBA with few alpha levels, then INDEX back to them: alpha travels through the colour table
    out = bytearray()
    while len(out) < nbytes:
        if rng.random() < 0.3:
            out += bytes([0xFF, int(rng.integers(0, 256)), int(rng.integers(0, 4)), int(rng.integers(0, 256)), int(rng.choice([0, 128, 255]))])
        elif rng.random() < 0.5:
            out.append(int(rng.integers(0, 64)))
        else:
            out += bytes([0xFE, int(rng.integers(0, 256)), int(rng.integers(0, 256)), int(rng.integers(0, 4))])
    return bytes(out[:nbytes])


def random_stream(rng: np.random.Generator, w: int, h: int, ch: int, ref) -> bytes:
    from fuzz_encode import random_image
    npx = w * h
    body = bytearray()
    target = int(npx * rng.choice([0.02, 0.3, 1.0, 1.3, 2.5, 5.5]))       # bytes of chunks: from far too few to more than any image needs
    target = max(1, min(target, 600_000))
    if rng.random() < 0.5:                                                # an encoder-made front (the speculation holds), hostile bytes behind it
        img = np.ascontiguousarray(random_image(rng, w, h)[:, :, :ch])
        enc = ref.encode(img, w, h, ch)
        cut = int(rng.integers(0, max(1, len(enc) - 22)))
        body += enc[14:14 + cut]
    while len(body) < target:
        body += soup(rng, int(rng.choice([1, 5, 40, 130, 700, 4096, 5000, 20000])))
    body = body[:target]
    hdr = b"qoif" + struct.pack(">II", w, h) + bytes([ch, int(rng.integers(0, 2))])
    return hdr + bytes(body) + END


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200, help="calls of qoimi_decode_batch (1..16 streams each)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--keep-going", action="store_true")
    args = ap.parse_args()
    import torch
    from oracle import oracle_py
    from qoi_amd import api
    ref = oracle_py.load_ref() or oracle_py.load_port()
    rng = np.random.default_rng(args.seed)
    images = failures = 0
    rounds_max = 0
    t0 = time.time()
    for it in range(args.iters):
        mode = int(rng.integers(0, 3))
        if mode == 0:
            w, h = int(rng.integers(1, 90)), int(rng.integers(1, 90))
        elif mode == 1:
            w, h = int(rng.integers(100, 1500)), int(rng.integers(50, 700))
        else:
            w, h = int(rng.integers(1, 30)), int(rng.integers(2000, 30000))
        ch = int(rng.choice([3, 4]))
        och = int(rng.choice([0, 3, 4]))
        n = int(rng.choice([1, 1, 2, 5, 16]))
        seg = str(rng.choice(["", "", "128", "256", "512", "1024", "2048", "4096", "320"]))
        if seg:
            os.environ["QOIMI_SEG_BYTES"] = seg
        else:
            os.environ.pop("QOIMI_SEG_BYTES", None)
        streams = [random_stream(rng, w, h, ch, ref) for _ in range(n)]
        c = api.Context(0)
        oc = och if och else ch
        sstride = (max(len(s) for s in streams) + 64 + 255) // 256 * 256
        pstride = (w * h * oc + 255) // 256 * 256
        d_s = torch.zeros(n * sstride, dtype=torch.uint8, device="cuda")
        for i, s in enumerate(streams):
            d_s[i * sstride:i * sstride + len(s)].copy_(torch.from_numpy(np.frombuffer(s, dtype=np.uint8).copy()))
        d_p = torch.full((n * pstride,), 0xCD, dtype=torch.uint8, device="cuda")
        desc = api.QoiDesc(w, h, ch, 0)
        c.decode_batch(d_s.data_ptr(), sstride, [len(s) for s in streams], [desc] * n, och, d_p.data_ptr(), pstride, torch.cuda.current_stream().cuda_stream)
        got = d_p.cpu().numpy()
        rounds_max = max(rounds_max, int(c.decode_stats()["rounds"]))
        for i, s in enumerate(streams):
            want, _ = ref.decode(s, och)
            if want is None or not np.array_equal(got[i * pstride:i * pstride + w * h * oc], want):
                bad = -1 if want is None else int(np.argmax(got[i * pstride:i * pstride + w * h * oc] != want))
                path = f"/tmp/fuzz_decode_batch_fail_{args.seed}_{it}_{i}.qoi"
                open(path, "wb").write(s)
                print(f"MISMATCH iter {it} stream {i}: {w}x{h}x{ch} -> {oc} channels, {len(s)} bytes, batch {n}, segment '{seg}': first differing byte {bad}; stream saved to {path}")
                failures += 1
                if not args.keep_going:
                    return 1
        images += n
        c.close()
        del d_s, d_p
    print(f"fuzz_decode_batch: {args.iters} calls, {images} streams, seed {args.seed}: every image equal to the {ref.kind} decoder's; most repair rounds in a call {rounds_max}; "
          f"{time.time() - t0:.0f} s" + (f"; {failures} MISMATCHES" if failures else ""))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
