#!/usr/bin/env python
"""qoibench for the MI355X path — same measurement semantics and table as the
reference's qoibench.c, with the GPU library in the `qoi` row's place.

    python tools/qoibench_mi355x.py <iterations> <directory> [options]

What is kept from qoibench.c: the directory walk (recursive unless --norecurse, one table per directory and a
grand total, `qoibench.c:491-555`), the round-trip verification of every image before timing
(`qoibench.c:408-417`, off with --noverify), BENCHMARK_FN's timing (`qoibench.c:364-376`: one discarded warm-up
unless --nowarmup, mean of N runs, allocation of the result inside the timed region), the flags --nowarmup
--noverify --noencode --nodecode --norecurse --onlytotals, and the output table
(`qoibench.c:335-360`: decode ms | encode ms | decode mpps | encode mpps | size kb | rate).

What differs: images are read from `*.qoi` files (decoded once with the oracle to obtain the raw pixels) instead
of `*.png` — PNG decoding is the job of libpng/stb_image, third-party code the reference does not vendor and this
image does not have; the libpng/stbi comparison rows are therefore absent (qoibench's own --nopng).  Rows:

    qoi-ref:     the reference implementation on one host core (the oracle build; a baseline)
    qoi-mi355x:  qoi_encode()/qoi_decode() of libqoi_mi355x.so, host pointers in and out (the drop-in; pays PCIe)
    qoi-dev:     the same kernels on device-resident buffers (qoimi_encode_batch / qoimi_decode_batch)

--nogpu prints the reference row only (no MI355X needed); --synth K writes K synthetic .qoi files of every
content class into the directory first (there are no test images in the reference tree).
"""
from __future__ import annotations

import argparse
import os
import struct
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ROWS = ("qoi-ref:   ", "qoi-mi355x:", "qoi-dev:   ")


class Result:
    def __init__(self):
        self.count = 0; self.raw_size = 0; self.px = 0
        self.libs = [[0, 0, 0] for _ in ROWS]          # size, encode_ns, decode_ns

    def add(self, o: "Result"):
        self.count += o.count; self.raw_size += o.raw_size; self.px += o.px
        for a, b in zip(self.libs, o.libs):
            for k in range(3):
                a[k] += b[k]


def print_result(res: Result, rows) -> str:
    """qoibench.c:335-360, same columns and formats."""
    px = res.px / res.count
    raw = res.raw_size / res.count
    lines = ["          decode ms   encode ms   decode mpps   encode mpps   size kb    rate"]
    for i in rows:
        size, enc, dec = (v / res.count for v in res.libs[i])
        lines.append("%s   %8.1f    %8.1f      %8.2f      %8.2f  %8d   %4.1f%%" % (
            ROWS[i], dec / 1e6, enc / 1e6, (px / (dec / 1000.0) if dec > 0 else 0), (px / (enc / 1000.0) if enc > 0 else 0),
            int(size) // 1024, size / raw * 100.0))
    lines.append("")
    return "\n".join(lines)


def bench_fn(nowarmup: bool, runs: int, fn) -> int:
    """BENCHMARK_FN (qoibench.c:364-376): mean ns over `runs`, the first run ignored unless nowarmup."""
    total = 0
    for i in range(1 if nowarmup else 0, runs + 1):
        t0 = time.perf_counter_ns()
        fn()
        t1 = time.perf_counter_ns()
        if i > 0:
            total += t1 - t0
    return total // runs


def benchmark_image(path: str, opt, ref, gpu) -> Result:
    stream_in = open(path, "rb").read()
    pixels, d = ref.decode(stream_in, 0)
    if pixels is None:
        raise SystemExit(f"Error decoding {path}")
    w, h, ch = d.width, d.height, d.channels
    res = Result(); res.count = 1; res.raw_size = w * h * ch; res.px = w * h
    from oracle import oracle_py
    desc_ref = oracle_py.QoiDesc(w, h, ch, 0)
    encoded = ref.encode(pixels, w, h, ch)
    if gpu:
        api, ctx, torch = gpu
        desc = api.QoiDesc(w, h, ch, api.QOI_SRGB)
        enc_gpu = api.qoi_encode(pixels, desc)
        if enc_gpu is None:
            raise SystemExit(f"Error encoding {path}")
        if not opt.noverify:                                       # qoibench.c:408-417
            back, _ = api.qoi_decode(enc_gpu, ch)
            if back is None or not np.array_equal(back, pixels):
                raise SystemExit(f"QOI roundtrip pixel mismatch for {path}")
    # row 0: reference on the host
    if not opt.nodecode:
        def f():
            p, _ = ref.decode_raw(buf_addr, len(encoded), 4); ref.free(p)
        import ctypes
        cbuf = (ctypes.c_ubyte * len(encoded)).from_buffer_copy(encoded); buf_addr = ctypes.addressof(cbuf)
        res.libs[0][2] = bench_fn(opt.nowarmup, opt.runs, f)
    if not opt.noencode:
        def f():
            p, n = ref.encode_raw(pixels.ctypes.data, desc_ref); ref.free(p)
        res.libs[0][1] = bench_fn(opt.nowarmup, opt.runs, f)
        res.libs[0][0] = len(encoded)
    if gpu:
        # row 1: drop-in, host pointers
        if not opt.nodecode:
            res.libs[1][2] = bench_fn(opt.nowarmup, opt.runs, lambda: api.qoi_decode(enc_gpu, 4))
        if not opt.noencode:
            res.libs[1][1] = bench_fn(opt.nowarmup, opt.runs, lambda: api.qoi_encode(pixels, desc))
            res.libs[1][0] = len(enc_gpu)
        # row 2: device-resident
        pstride = (w * h * 4 + 255) // 256 * 256
        sstride = (api.encode_bound(w, h, ch) + 255) // 256 * 256
        dpx = torch.from_numpy(pixels.copy()).cuda()
        dst = torch.empty(sstride, dtype=torch.uint8, device="cuda")
        dout = torch.empty(pstride, dtype=torch.uint8, device="cuda")
        dlen = torch.zeros(1, dtype=torch.int32, device="cuda")
        st = torch.cuda.current_stream().cuda_stream

        def enc():
            ctx.encode_batch(dpx.data_ptr(), dpx.numel(), desc, 1, dst.data_ptr(), sstride, dlen.data_ptr(), st)
            torch.cuda.synchronize()

        def dec():
            ctx.decode_batch(dst.data_ptr(), sstride, [len(enc_gpu)], [desc], 4, dout.data_ptr(), pstride, st)
        enc()
        if not opt.nodecode:
            res.libs[2][2] = bench_fn(opt.nowarmup, opt.runs, dec)
        if not opt.noencode:
            res.libs[2][1] = bench_fn(opt.nowarmup, opt.runs, enc)
            res.libs[2][0] = len(enc_gpu)
    return res


def benchmark_directory(path: str, grand: Result, opt, ref, gpu, rows, out):
    entries = sorted(os.listdir(path))
    if not opt.norecurse:                                            # qoibench.c:497-511
        for e in entries:
            sub = os.path.join(path, e)
            if os.path.isdir(sub) and not e.startswith("."):
                benchmark_directory(sub, grand, opt, ref, gpu, rows, out)
    dirtotal = Result()
    has_shown_head = False
    for e in entries:
        if not e.endswith(".qoi"):
            continue
        if not has_shown_head:
            has_shown_head = True
            out(f"## Benchmarking {path}/*.qoi -- {opt.runs} runs\n")
        f = os.path.join(path, e)
        res = benchmark_image(f, opt, ref, gpu)
        if not opt.onlytotals:
            w, h = struct.unpack(">II", open(f, "rb").read(12)[4:12])
            out(f"## {f} size: {w}x{h}")
            out(print_result(res, rows))
        dirtotal.add(res)
    if dirtotal.count > 0:
        out(f"## Total for {path}")
        out(print_result(dirtotal, rows))
        grand.add(dirtotal)


def write_synth(directory: str, k: int):
    from oracle import oracle_py
    from qoi_amd import synth
    port = oracle_py.load_port()
    os.makedirs(directory, exist_ok=True)
    for kind in synth.KINDS:
        for i in range(k):
            w, h = (1920, 1080) if i % 2 == 0 else (640, 480)
            s = port.encode(synth.frame_rgba(kind, w, h, i), w, h, 4)
            open(os.path.join(directory, f"{kind}_{i}.qoi"), "wb").write(s)


def main(argv=None, out=print) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("runs", type=int)
    ap.add_argument("directory")
    for f in ("nowarmup", "noverify", "noencode", "nodecode", "norecurse", "onlytotals", "nogpu"):
        ap.add_argument("--" + f, action="store_true")
    ap.add_argument("--synth", type=int, default=0)
    opt = ap.parse_args(argv)
    if opt.runs < 1:
        raise SystemExit("Invalid number of runs")                  # qoibench.c:598-600
    if opt.synth:
        write_synth(opt.directory, opt.synth)
    from oracle import oracle_py
    ref = oracle_py.load_ref() or oracle_py.load_port()
    gpu = None
    rows = (0,)
    if not opt.nogpu:
        import torch
        from qoi_amd import api
        gpu = (api, api.Context(0), torch)
        rows = (0, 1, 2)
    grand = Result()
    benchmark_directory(opt.directory, grand, opt, ref, gpu, rows, out)
    if grand.count > 0:
        out("# Grand total for " + opt.directory)
        out(print_result(grand, rows))
    else:
        out(f"No images found in {opt.directory}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
