#!/usr/bin/env python
"""Differential fuzz harness for the GPU ENCODER (companion of tests/fuzz_decode.py).

Random images of random shapes, channel counts and batch sizes go through qoimi_encode_batch; every stream must be
byte-identical to what the reference encoder (qoi.h:356-486; the unmodified reference where oracle/_ref is built, else the C
restatement) writes for the same pixels, and decode back to the pixels through qoimi_decode_batch.  The images are made to
exercise what the set-parallel encoder cuts across: runs that cross step / group / slab / set boundaries and the 62 cap
(qoi.h:417-421), palettes whose colours share hash slots (qoi.h:430-436), alpha steps (qoi.h:461-474), stretches of noise that
spill a set's bytes, and flat content that takes the generic entry-state path.  Batch sizes 1..300 cover the three placement
forms (tree, look-back, order-free by QOIMI_ENC_LOOKBACK).

    python tests/fuzz_encode.py --iters 300 --seed 1            # needs an MI355X
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

os.environ.setdefault("QOIMI_TUNING", "1")      # the placement / segment-size knobs below are looked at only under it (qoi_host.hip)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def random_image(rng: np.random.Generator, w: int, h: int) -> np.ndarray:
    """h x w x 4 uint8: a patchwork of stretches (along the scan order, as the encoder sees the pixels)."""
    n = w * h
    a = np.empty((n, 4), dtype=np.uint8)
    palette = rng.integers(0, 256, size=(int(rng.integers(2, 40)), 4), dtype=np.uint8)
    if rng.random() < 0.5:
        palette[:, 3] = 255
    elif rng.random() < 0.5:
        palette[:, 3] = rng.choice([0, 128, 255], size=len(palette))
    pos = 0
    cur = palette[0].astype(np.int64)
    while pos < n:
        kind = int(rng.integers(0, 10))
        L = int(rng.choice([1, 2, 3, 5, 61, 62, 63, 64, 65, 124, 125, 500, 1024, 1025, 3071, 3072, 3100, 9000, 40000]))
        L = max(1, min(n - pos, int(L * rng.uniform(0.5, 1.5)) if rng.random() < 0.5 else L))
        seg = a[pos:pos + L]
        if kind == 0:                                             # one colour: a run (crosses whatever boundary lies in it)
            seg[:] = palette[int(rng.integers(0, len(palette)))]
        elif kind == 1:                                           # palette tiles of random width: INDEX hits and slot collisions
            tw = int(rng.integers(1, 200))
            idx = rng.integers(0, len(palette), size=L // tw + 2)
            seg[:] = palette[np.repeat(idx, tw)[:L]]
        elif kind == 2:                                           # noise (5-byte / 4-byte chunks: sets spill)
            seg[:] = rng.integers(0, 256, size=(L, 4), dtype=np.uint8)
            if rng.random() < 0.5:
                seg[:, 3] = cur[3]
        elif kind == 3:                                           # small steps: DIFF / LUMA with wrap-around
            d = rng.integers(-3, 4, size=(L, 3))
            if rng.random() < 0.5:
                d[:, 1] = rng.integers(-33, 34, size=L)
                d[:, 0] = d[:, 1] + rng.integers(-9, 9, size=L)
                d[:, 2] = d[:, 1] + rng.integers(-9, 9, size=L)
            seg[:, :3] = (cur[:3] + np.cumsum(d, axis=0)) & 255
            seg[:, 3] = cur[3]
        elif kind == 4:                                           # gradient with repeats: short runs between small deltas
            rep = int(rng.integers(1, 9))
            base = (cur[:3] + np.cumsum(rng.integers(-2, 2, size=(L // rep + 2, 3)), axis=0)) & 255
            seg[:, :3] = np.repeat(base, rep, axis=0)[:L]
            seg[:, 3] = cur[3]
        elif kind == 5:                                           # alpha steps on an otherwise smooth stretch
            seg[:, :3] = (cur[:3] + np.cumsum(rng.integers(-1, 2, size=(L, 3)), axis=0)) & 255
            seg[:, 3] = np.repeat(rng.integers(0, 256, size=L // 7 + 2), 7)[:L]
        elif kind == 6:                                           # two colours alternating: INDEX every pixel
            c = palette[rng.integers(0, len(palette), size=2)]
            seg[:] = c[np.arange(L) & 1]
        elif kind == 7:                                           # colours that share ONE hash slot, in short tiles: every change evicts (qoi.h:430-436)
            k = int(rng.integers(2, 6))
            c = rng.integers(0, 256, size=(k, 4), dtype=np.int64)
            if rng.random() < 0.5:
                c[:, 3] = cur[3]
            slot = int(rng.integers(0, 64))
            c[:, 0] = ((slot - 5 * c[:, 1] - 7 * c[:, 2] - 11 * c[:, 3]) * 43 % 64) + 64 * rng.integers(0, 4, size=k)   # 3 * 43 = 1 (mod 64)
            tw = int(rng.integers(1, 6))
            seg[:] = c[np.repeat(rng.integers(0, k, size=L // tw + 2), tw)[:L]].astype(np.uint8)
        elif kind == 8:                                           # all-zero pixels: equal to the zeroed table's words (qoi.h:393), INDEX 0 without ever having been stored
            seg[:] = palette[np.repeat(rng.integers(0, len(palette), size=L // 9 + 2), 9)[:L]]
            seg[rng.random(L) < 0.2] = 0
            if rng.random() < 0.3:
                seg[: L // 2] = 0
        else:                                                     # the start value and its neighbours (qoi.h:396-399: {0,0,0,255} is never in the table)
            seg[:] = np.array([0, 0, 0, 255], dtype=np.uint8)
            if L > 4 and rng.random() < 0.5:
                seg[L // 2] = (0, 0, 0, 0)
        cur = seg[-1].astype(np.int64)
        pos += L
    return a.reshape(h, w, 4)


def random_shape(rng: np.random.Generator, max_px: int):
    mode = int(rng.integers(0, 5))
    if mode == 0:
        w, h = int(rng.integers(1, 70)), int(rng.integers(1, 70))
    elif mode == 1:                                               # around slab / set multiples (1024, 3072 pixels)
        px = int(rng.choice([1023, 1024, 1025, 2048, 3071, 3072, 3073, 6144, 9216, 65536, 65537])) + int(rng.integers(-2, 3))
        w = int(rng.choice([1, 2, 7, 64, 511, 1024])); h = max(1, px // w)
    elif mode == 2:
        w, h = int(rng.integers(200, 2000)), int(rng.integers(100, 1200))
    elif mode == 3:                                               # very wide / very tall
        w, h = (int(rng.integers(3000, 20000)), int(rng.integers(1, 40))) if rng.random() < 0.5 else (int(rng.integers(1, 40)), int(rng.integers(3000, 20000)))
    else:
        w, h = int(rng.integers(1500, 4200)), int(rng.integers(900, 2400))
    if max_px >= 30_000_000 and rng.random() < 0.35:              # (--max-pixels 45000000: images past 12288 sets, which are placed order-free by default)
        w, h = int(rng.integers(5000, 9000)), int(rng.integers(3500, 6000))
    while w * h > max_px:
        h = max(1, h // 2)
    return w, h


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=200, help="calls of qoimi_encode_batch (1..300 images each)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-pixels", type=int, default=6_000_000)
    ap.add_argument("--only", type=int, default=-1, help="replay: run this iteration alone (the random sequence of the others is drawn and discarded)")
    ap.add_argument("--form", default=None, help="replay: override QOIMI_ENC_LOOKBACK ('' = the library's choice)")
    ap.add_argument("--slabs", default=None, help="replay: override QOIMI_ENC_SET_SLABS")
    ap.add_argument("--keep-going", action="store_true", help="report every mismatch instead of stopping at the first")
    ap.add_argument("--odd-strides", action="store_true", help="device buffers at odd addresses with odd strides; also checks that nothing is written outside a stream / an image")
    ap.add_argument("--seconds", type=float, default=0.0, help="time budget: stop after the call during which this many seconds have passed (with --iters as the upper bound)")
    ap.add_argument("--batch8-half", action="store_true", help="half of the calls carry 8 or more images (shapes shrunk to fit): look-back placement with tickets and "
                    "SPREAD - the path of the benchmark's batches")
    ap.add_argument("--dropin", action="store_true", help="drive the drop-in qoi_encode / qoi_decode on host pointers instead (one image per call, sizes jumping "
                    "up and down: the result buffer of qoi_encode is sized by the thread's previous stream)")
    args = ap.parse_args()
    import torch
    from gpu_util import DeviceBatch
    from oracle import oracle_py
    from qoi_amd import api
    ref = oracle_py.load_ref() or oracle_py.load_port()
    rng = np.random.default_rng(args.seed)
    images = px_total = failures = 0
    t0 = time.time()
    if args.dropin:
        for it in range(args.iters):
            w, h = random_shape(rng, args.max_pixels)
            ch = int(rng.choice([3, 4]))
            f = np.ascontiguousarray(random_image(rng, w, h)[:, :, :ch])
            if rng.random() < 0.25:                                   # all noise: the longest stream right behind a short one
                f = rng.integers(0, 256, size=f.shape, dtype=np.uint8)
            got = api.qoi_encode(f, api.QoiDesc(w, h, ch, int(rng.integers(0, 2))))
            want = ref.encode(f, w, h, ch)
            ok = got is not None and got[:12] == want[:12] and got[14:] == want[14:]      # (byte 13 = the colorspace handed in)
            och = int(rng.choice([0, 3, 4]))
            back, d = api.qoi_decode(want, och) if ok else (None, None)
            wpx, _ = ref.decode(want, och)
            if not ok or back is None or not np.array_equal(back, wpx):
                print(f"DROP-IN MISMATCH iter {it}: {w}x{h}x{ch} -> {och}: encode {'ok' if ok else 'differs'}")
                failures += 1
                if not args.keep_going:
                    return 1
            images += 1
            px_total += w * h
        print(f"fuzz_encode --dropin: {images} qoi_encode + qoi_decode calls on host pointers, {px_total / 1e6:.0f} Mpx, seed {args.seed}: every stream byte-identical to the "
              f"{ref.kind} encoder's, every decode equal to its decoder's; {time.time() - t0:.0f} s" + (f"; {failures} MISMATCHES" if failures else ""))
        return 1 if failures else 0
    forms = {"": 0, "0": 0, "1": 0, "2": 0}
    calls_done = 0
    for it in range(args.iters):
        w, h = random_shape(rng, args.max_pixels)
        ch = int(rng.choice([3, 4]))
        n = int(rng.choice([1, 1, 1, 2, 3, 7, 8, 9, 12, 40, 150, 300]))      # (the large counts only stay large for small shapes)
        while n > 1 and n * w * h > 2 * args.max_pixels:
            n -= 1
        if args.batch8_half and rng.random() < 0.5 and n < 8:
            n = int(rng.choice([8, 9, 12, 16, 33, 64]))
            while n * w * h > 2 * args.max_pixels and h > 1:
                h = max(1, h // 2)
            while n * w * h > 2 * args.max_pixels and w > 1:
                w = max(1, w // 2)
        form = str(rng.choice(["", "", "", "0", "1", "2"]))           # mostly the library's own choice
        slabs = str(rng.choice(["", "", "1", "2", "3", "5"]))
        for k, v in (("QOIMI_ENC_LOOKBACK", form), ("QOIMI_ENC_SET_SLABS", slabs)):
            if v:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        frames = [np.ascontiguousarray(random_image(rng, w, h)[:, :, :ch]) for _ in range(n)]
        if args.only >= 0 and it != args.only:
            continue
        if args.form is not None:
            form = args.form
        if args.slabs is not None:
            slabs = args.slabs
        for k, v in (("QOIMI_ENC_LOOKBACK", form), ("QOIMI_ENC_SET_SLABS", slabs)):
            if v:
                os.environ[k] = v
            else:
                os.environ.pop(k, None)
        forms[form] += 1
        c = api.Context(0)
        if args.odd_strides:
            # buffers at odd addresses with odd strides: the C-ABI states no alignment for d_pixels / d_streams or their strides
            npx = w * h
            desc = api.QoiDesc(w, h, ch, 0)
            po, so, oo = (int(x) for x in rng.integers(0, 16, size=3))
            ps = npx * ch + int(rng.integers(0, 19)); ss = api.encode_bound(w, h, ch) + int(rng.integers(0, 35)); ost = npx * ch + int(rng.integers(0, 19))
            d_pix = torch.zeros(po + n * ps + 64, dtype=torch.uint8, device="cuda")
            d_str = torch.full((so + n * ss + 64,), 0xEE, dtype=torch.uint8, device="cuda")
            d_out = torch.full((oo + n * ost + 64,), 0xCD, dtype=torch.uint8, device="cuda")
            d_len = torch.zeros(n, dtype=torch.int32, device="cuda")
            for i, f in enumerate(frames):
                d_pix[po + i * ps:po + i * ps + npx * ch].copy_(torch.from_numpy(f.reshape(-1)))
            st = torch.cuda.current_stream().cuda_stream
            c.encode_batch(d_pix.data_ptr() + po, ps, desc, n, d_str.data_ptr() + so, ss, d_len.data_ptr(), st)
            c.encode_status(st)
            lens = d_len.cpu().numpy()
            hs = d_str.cpu().numpy()
            ok = True
            for i, f in enumerate(frames):
                want = np.frombuffer(ref.encode(f, w, h, ch), dtype=np.uint8)
                got = hs[so + i * ss:so + i * ss + int(lens[i])]
                guard = hs[so + i * ss + int(lens[i]):so + (i + 1) * ss]          # nothing written behind a stream (up to the next one)
                if len(got) != len(want) or not np.array_equal(got, want) or (len(want) < ss and not (guard == 0xEE).all()):
                    print(f"MISMATCH (odd strides) iter {it} image {i}: {w}x{h}x{ch}, batch {n}, form '{form}', slabs '{slabs}', offsets {po} {so}, strides {ps} {ss}: "
                          f"{len(got)} bytes against {len(want)}, bytes behind the stream untouched: {bool((guard == 0xEE).all())}")
                    ok = False
            c.decode_batch(d_str.data_ptr() + so, ss, [int(x) for x in lens], [desc] * n, ch, d_out.data_ptr() + oo, ost, st)
            ho = d_out.cpu().numpy()
            for i, f in enumerate(frames):
                if not np.array_equal(ho[oo + i * ost:oo + i * ost + npx * ch], f.reshape(-1)) or not (ho[oo + i * ost + npx * ch:oo + (i + 1) * ost] == 0xCD).all():
                    print(f"ROUND TRIP MISMATCH (odd strides) iter {it} image {i}: {w}x{h}x{ch}, batch {n}, offset {oo}, stride {ost}")
                    ok = False
            if not (ho[:oo] == 0xCD).all() or not (hs[:so] == 0xEE).all():
                print(f"WRITE IN FRONT OF A BUFFER iter {it}")
                ok = False
            if not ok:
                failures += 1
                if not args.keep_going:
                    return 1
            images += n
            px_total += n * npx
            c.close()
            del d_pix, d_str, d_out
            continue
        b = DeviceBatch(c, w, h, ch, n)
        for i, f in enumerate(frames):
            b.upload(i, f)
        lens = b.encode()
        torch.cuda.synchronize()
        for i, f in enumerate(frames):
            want = ref.encode(f, w, h, ch)
            got = b.stream_bytes(i, lens[i])
            if got != want:
                path = f"/tmp/fuzz_encode_fail_{args.seed}_{it}_{i}.npy"
                np.save(path, f)
                g8, w8 = np.frombuffer(got, dtype=np.uint8), np.frombuffer(want, dtype=np.uint8)
                m = min(len(g8), len(w8))
                first = int(np.argmax(g8[:m] != w8[:m])) if (g8[:m] != w8[:m]).any() else m
                print(f"MISMATCH iter {it} image {i}: {w}x{h}x{ch}, batch {n}, form '{form}', slabs '{slabs}': {len(got)} bytes against {len(want)}, first difference at byte {first}"
                      f" (got {g8[first:first + 8].tolist()} want {w8[first:first + 8].tolist()}); pixels saved to {path}")
                failures += 1
                if not args.keep_going:
                    return 1
        out = torch.full((n * b.pixel_stride,), 0xCD, dtype=torch.uint8, device="cuda")
        stride = b.decode_into(out, lens, ch)
        got = out.cpu().numpy()
        for i, f in enumerate(frames):
            if not np.array_equal(got[i * stride:i * stride + w * h * ch], f.reshape(-1)):
                print(f"ROUND TRIP MISMATCH iter {it} image {i}: {w}x{h}x{ch}, batch {n}")
                return 1
        images += n
        px_total += n * w * h
        c.close()
        del b, out
        calls_done = it + 1
        if args.seconds > 0 and time.time() - t0 > args.seconds:
            break
    else:
        calls_done = args.iters
    print(f"fuzz_encode: {calls_done} calls, {images} images, {px_total / 1e6:.0f} Mpx, seed {args.seed}: every stream byte-identical to the {ref.kind} encoder's, "
          f"every round trip exact; placement forced order-free / look-back / tree in {forms['0']} / {forms['1']} / {forms['2']} calls, the library's choice in {forms['']}; "
          f"{time.time() - t0:.0f} s" + (f"; {failures} MISMATCHES" if failures else ""))
    return 1 if failures else 0


if __name__ == "__main__":
    sys.exit(main())
