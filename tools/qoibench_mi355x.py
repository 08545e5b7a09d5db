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

What differs: PNGs are read with `tools/png_io.py` (the reference uses stb_image / libpng, third-party code it does
not vendor and this image does not have); the libpng/stbi comparison rows are absent (qoibench's own --nopng).
Rows:

    qoi-mi355x:  qoi_encode()/qoi_decode() of libqoi_mi355x.so, host pointers in and out (the drop-in; pays PCIe)
    qoi-dev:     the same kernels on device-resident buffers (qoimi_encode_batch / qoimi_decode_batch)
    qoi-ref:     only with --ref-lib PATH [--ref-prefix P]: a CPU build of the reference's qoi_encode/qoi_decode
                 (any shared object exporting `<P>qoi_encode` / `<P>qoi_decode`) timed beside, one host core
    qoi-batch:   (totals only) ALL images of a directory in ONE qoimi_encode_images call and ONE qoimi_decode_batch call,
                 device-resident - whatever their shapes and channel counts; per-image means like the other rows; --nobatch: off

--synth K writes K synthetic .png files of every content class into the directory first (there are no test images
in the reference tree).  The tool itself contains no CPU codec and needs the MI355X for its own rows.
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import ctypes

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import png_io  # noqa: E402

ROWS = ("qoi-mi355x:", "qoi-dev:   ", "qoi-ref:   ", "qoi-batch: ")


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


class RefCodec:
    """A CPU build of the reference ABI (qoi.h:278, 289) loaded from a shared object - the optional baseline row."""

    class Desc(ctypes.Structure):
        _fields_ = [("width", ctypes.c_uint), ("height", ctypes.c_uint), ("channels", ctypes.c_ubyte), ("colorspace", ctypes.c_ubyte)]

    def __init__(self, path: str, prefix: str = ""):
        lib = ctypes.CDLL(path)
        self.enc = getattr(lib, prefix + "qoi_encode"); self.dec = getattr(lib, prefix + "qoi_decode")
        self.enc.restype = ctypes.c_void_p
        self.enc.argtypes = [ctypes.c_void_p, ctypes.POINTER(self.Desc), ctypes.POINTER(ctypes.c_int)]
        self.dec.restype = ctypes.c_void_p
        self.dec.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(self.Desc), ctypes.c_int]
        self.free = ctypes.CDLL(None).free
        self.free.argtypes = [ctypes.c_void_p]; self.free.restype = None

    def encode(self, pixels: np.ndarray, w: int, h: int, ch: int):
        d = self.Desc(w, h, ch, 0); n = ctypes.c_int(0)
        p = self.enc(pixels.ctypes.data, ctypes.byref(d), ctypes.byref(n))
        return p, n.value


def load_image(path: str):
    """uint8[h, w, ch] like qoibench.c:389-398: the PNG's own 3 channels, anything else as RGBA."""
    data = open(path, "rb").read()
    w, h, ch = png_io.png_info(data)
    ch = 3 if ch == 3 else 4
    px, w, h = png_io.read_png(data, ch)
    return np.ascontiguousarray(px), w, h, ch


def benchmark_image(path: str, opt, ref, gpu) -> Result:
    try:
        pixels, w, h, ch = load_image(path)
    except (OSError, png_io.PngError):
        raise SystemExit(f"Error decoding header {path}")             # qoibench.c:385-388
    res = Result(); res.count = 1; res.raw_size = w * h * ch; res.px = w * h
    if gpu:
        api, ctx, torch = gpu
        desc = api.QoiDesc(w, h, ch, api.QOI_SRGB)
        enc_gpu = api.qoi_encode(pixels, desc)
        if enc_gpu is None:
            raise SystemExit(f"Error encoding {path}")
        if not opt.noverify:                                       # qoibench.c:408-417
            back, _ = api.qoi_decode(enc_gpu, ch)
            if back is None or not np.array_equal(back, pixels.reshape(-1)):
                raise SystemExit(f"QOI roundtrip pixel mismatch for {path}")
        # row 0: drop-in, host pointers
        if not opt.nodecode:
            res.libs[0][2] = bench_fn(opt.nowarmup, opt.runs, lambda: api.qoi_decode(enc_gpu, 4))
        if not opt.noencode:
            res.libs[0][1] = bench_fn(opt.nowarmup, opt.runs, lambda: api.qoi_encode(pixels, desc))
            res.libs[0][0] = len(enc_gpu)
        # row 1: device-resident
        pstride = (w * h * 4 + 255) // 256 * 256
        sstride = (api.encode_bound(w, h, ch) + 255) // 256 * 256
        dpx = torch.from_numpy(pixels.reshape(-1).copy()).cuda()
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
            res.libs[1][2] = bench_fn(opt.nowarmup, opt.runs, dec)
        if not opt.noencode:
            res.libs[1][1] = bench_fn(opt.nowarmup, opt.runs, enc)
            res.libs[1][0] = len(enc_gpu)
    if ref:
        # row 2: a CPU build of the reference on the host (baseline; malloc/free inside the timed region)
        p, n = ref.encode(pixels, w, h, ch)
        if not p:
            raise SystemExit(f"Error encoding {path}")
        encoded = ctypes.string_at(p, n); ref.free(p)
        cbuf = (ctypes.c_ubyte * n).from_buffer_copy(encoded); buf_addr = ctypes.addressof(cbuf)
        if not opt.nodecode:
            def f():
                d = RefCodec.Desc(); q = ref.dec(buf_addr, n, ctypes.byref(d), 4); ref.free(q)
            res.libs[2][2] = bench_fn(opt.nowarmup, opt.runs, f)
        if not opt.noencode:
            def f():
                q, _ = ref.encode(pixels, w, h, ch); ref.free(q)
            res.libs[2][1] = bench_fn(opt.nowarmup, opt.runs, f)
            res.libs[2][0] = n
    return res


def benchmark_batch(images, streams, opt, gpu) -> list:
    """The directory's images in one qoimi_encode_images call and one qoimi_decode_batch call (device-resident):
    [stream bytes, encode ns, decode ns] for the whole directory.  `streams`: what the drop-in qoi_encode wrote for each image -
    the batch call must write the same bytes."""
    api, ctx, torch = gpu
    descs = [api.QoiDesc(w, h, ch, api.QOI_SRGB) for (_, w, h, ch) in images]
    pix_off, str_off = [], []
    po = so = 0
    for (px, w, h, ch) in images:
        pix_off.append(po); str_off.append(so)
        po += (px.size + 255) // 256 * 256; so += (api.encode_bound(w, h, ch) + 255) // 256 * 256
    d_pix = torch.empty(po, dtype=torch.uint8, device="cuda")
    d_str = torch.empty(so, dtype=torch.uint8, device="cuda")
    d_len = torch.zeros(len(images), dtype=torch.int32, device="cuda")
    for (px, _, _, _), o in zip(images, pix_off):
        d_pix[o:o + px.size].copy_(torch.from_numpy(px.reshape(-1)))
    st = torch.cuda.current_stream().cuda_stream

    def enc():
        ctx.encode_images(d_pix.data_ptr(), pix_off, descs, d_str.data_ptr(), str_off, d_len.data_ptr(), st)
        ctx.encode_status(st)
    enc()
    lens = [int(x) for x in d_len.cpu().numpy()]
    if not opt.noverify:
        for i, want in enumerate(streams):
            if d_str[str_off[i]:str_off[i] + lens[i]].cpu().numpy().tobytes() != want:
                raise SystemExit(f"batch stream {i} differs from the drop-in's")
    # decode: the streams at a common stride, all of them to 4 channels (qoibench.c decodes with channels = 4)
    sstride = (max(lens) + 8 + 255) // 256 * 256
    pstride = (max(w * h for (_, w, h, _) in images) * 4 + 255) // 256 * 256
    d_in = torch.zeros(len(images) * sstride, dtype=torch.uint8, device="cuda")
    for i in range(len(images)):
        d_in[i * sstride:i * sstride + lens[i]].copy_(d_str[str_off[i]:str_off[i] + lens[i]])
    d_out = torch.empty(len(images) * pstride, dtype=torch.uint8, device="cuda")

    def dec():
        ctx.decode_batch(d_in.data_ptr(), sstride, lens, descs, 4, d_out.data_ptr(), pstride, st)
    row = [sum(lens), 0, 0]
    if not opt.nodecode:
        row[2] = bench_fn(opt.nowarmup, opt.runs, dec)
    if not opt.noencode:
        row[1] = bench_fn(opt.nowarmup, opt.runs, enc)
    return row


def benchmark_directory(path: str, grand: Result, opt, ref, gpu, rows, out):
    entries = sorted(os.listdir(path))
    if not opt.norecurse:                                            # qoibench.c:497-511
        for e in entries:
            sub = os.path.join(path, e)
            if os.path.isdir(sub) and not e.startswith("."):
                benchmark_directory(sub, grand, opt, ref, gpu, rows, out)
    dirtotal = Result()
    has_shown_head = False
    batch_images, batch_streams = [], []
    for e in entries:
        if not e.endswith(".png"):
            continue
        if not has_shown_head:
            has_shown_head = True
            out(f"## Benchmarking {path}/*.png -- {opt.runs} runs\n")
        f = os.path.join(path, e)
        res = benchmark_image(f, opt, ref, gpu)
        if not opt.onlytotals:
            w, h, _ = png_io.png_info(open(f, "rb").read())
            out(f"## {f} size: {w}x{h}")
            out(print_result(res, [r for r in rows if r != 3]))
        dirtotal.add(res)
        if gpu and 3 in rows:
            px, w, h, ch = load_image(f)
            batch_images.append((px, w, h, ch))
            batch_streams.append(gpu[0].qoi_encode(px, gpu[0].QoiDesc(w, h, ch, gpu[0].QOI_SRGB)))
    if dirtotal.count > 0:
        if batch_images:
            dirtotal.libs[3] = benchmark_batch(batch_images, batch_streams, opt, gpu)
        out(f"## Total for {path}")
        out(print_result(dirtotal, rows))
        grand.add(dirtotal)


def write_synth(directory: str, k: int):
    from qoi_amd import synth
    os.makedirs(directory, exist_ok=True)
    for kind in synth.KINDS:
        for i in range(k):
            w, h = (1920, 1080) if i % 2 == 0 else (640, 480)
            open(os.path.join(directory, f"{kind}_{i}.png"), "wb").write(png_io.write_png(synth.frame_rgba(kind, w, h, i), 1))


def main(argv=None, out=print, ref=None, use_gpu=True) -> int:
    """`ref` / `use_gpu=False`: the tests drive the table and flag logic with a CPU reference build and no GPU."""
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("runs", type=int)
    ap.add_argument("directory")
    for f in ("nowarmup", "noverify", "noencode", "nodecode", "norecurse", "onlytotals"):
        ap.add_argument("--" + f, action="store_true")
    ap.add_argument("--nobatch", action="store_true", help="no qoi-batch row (a directory's images in one call)")
    ap.add_argument("--synth", type=int, default=0)
    ap.add_argument("--ref-lib", default="", help="shared object with a CPU build of the reference ABI: adds the qoi-ref row")
    ap.add_argument("--ref-prefix", default="", help="symbol prefix in --ref-lib (e.g. ref_)")
    opt = ap.parse_args(argv)
    if opt.runs < 1:
        raise SystemExit("Invalid number of runs")                  # qoibench.c:598-600
    if opt.synth:
        write_synth(opt.directory, opt.synth)
    if ref is None and opt.ref_lib:
        ref = RefCodec(opt.ref_lib, opt.ref_prefix)
    gpu = None
    rows = ()
    if use_gpu:
        import torch
        from qoi_amd import api
        gpu = (api, api.Context(0), torch)                           # fails loudly without the library / a gfx950 GPU
        rows = (0, 1)
    if ref:
        rows = rows + (2,)
    if use_gpu and not opt.nobatch:
        rows = rows + (3,)
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
