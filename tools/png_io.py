"""Minimal PNG reader / writer for the conversion tool (host-side plumbing, numpy + zlib only).

The reference's ``qoiconv.c`` reads PNGs through stb_image (``qoiconv.c:21-24``) and writes them through
stb_image_write (``qoiconv.c:26-27``); both are third-party single headers that are not part of the
reference repository and are absent here.  This module restates just what the tool needs from the PNG
specification (ISO/IEC 15948): critical chunks IHDR / PLTE / IDAT / IEND plus tRNS, colour types 0, 2, 3, 4, 6,
bit depths 1-16 (16-bit samples keep their high byte, sub-byte samples are scaled like stb_image does),
the five scanline filters, non-interlaced images only.  Output is always 8-bit RGB or RGBA, as
``stbi_load(..., channels)`` would hand it to ``qoi_write`` (``qoiconv.c:49-55``).
"""
from __future__ import annotations

import struct
import zlib
from typing import Tuple

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"
_SAMPLES = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}


class PngError(ValueError):
    pass


def _chunks(data: bytes):
    if data[:8] != _SIG:
        raise PngError("not a PNG file")
    p = 8
    while p + 8 <= len(data):
        n, typ = struct.unpack(">I4s", data[p:p + 8])
        body = data[p + 8:p + 8 + n]
        if len(body) != n or p + 12 + n > len(data):
            raise PngError("truncated chunk")
        crc = struct.unpack(">I", data[p + 8 + n:p + 12 + n])[0]
        if zlib.crc32(typ + body) & 0xFFFFFFFF != crc:
            raise PngError(f"bad CRC in {typ!r}")
        yield typ, body
        p += 12 + n
        if typ == b"IEND":
            return
    raise PngError("no IEND")


def png_info(data: bytes) -> Tuple[int, int, int]:
    """(width, height, channels) like ``stbi_info``: channels 1..4 of the file's own format (tRNS adds alpha)."""
    w = h = ch = 0
    ctype = None
    for typ, body in _chunks(data):
        if typ == b"IHDR":
            w, h, depth, ctype, comp, flt, il = struct.unpack(">IIBBBBB", body)
            ch = {0: 1, 2: 3, 3: 3, 4: 2, 6: 4}.get(ctype, 0)
        elif typ == b"tRNS" and ctype in (0, 2, 3):
            ch = {0: 2, 2: 4, 3: 4}[ctype]
        elif typ == b"IDAT":
            break
    if not ch:
        raise PngError("no IHDR")
    return w, h, ch


_UNFILTER = None


def _native_unfilter():
    """tools/png_unfilter.c compiled with gcc on first use (None if there is no compiler)."""
    global _UNFILTER
    if _UNFILTER is None:
        import ctypes
        import os
        import subprocess
        import tempfile
        here = os.path.dirname(os.path.abspath(__file__))
        so = os.path.join(tempfile.gettempdir(), f"qoimi_png_unfilter_{os.getuid()}.so")
        try:
            src = os.path.join(here, "png_unfilter.c")
            if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
                subprocess.run(["gcc", "-O2", "-shared", "-fPIC", "-o", so, src], check=True, capture_output=True)
            lib = ctypes.CDLL(so)
            lib.png_unfilter.restype = ctypes.c_int
            lib.png_unfilter.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
            _UNFILTER = lib.png_unfilter
        except Exception:
            _UNFILTER = False
    return _UNFILTER or None


def _unfilter(raw: np.ndarray, h: int, stride: int, bpp: int) -> np.ndarray:
    fn = _native_unfilter()
    if fn is not None:
        src = np.ascontiguousarray(raw[:h * (stride + 1)])
        out = np.empty((h, stride), dtype=np.uint8)
        if fn(src.ctypes.data, out.ctypes.data, h, stride, bpp) != 0:
            raise PngError("bad filter type")
        return out
    return _unfilter_py(raw, h, stride, bpp)


def _unfilter_py(raw: np.ndarray, h: int, stride: int, bpp: int) -> np.ndarray:
    out = np.zeros((h + 1, stride), dtype=np.uint8)          # row 0: the all-zero "prior" of the first scanline
    pos = 0
    for y in range(1, h + 1):
        ft = int(raw[pos]); line = raw[pos + 1:pos + 1 + stride]; pos += 1 + stride
        prior = out[y - 1]
        if ft == 0:
            out[y] = line
        elif ft == 2:
            out[y] = line + prior
        elif ft == 1:                                          # Sub: running sum per byte lane
            cur = line.astype(np.uint32)
            lanes = cur.copy()
            for k in range(bpp):
                lanes[k::bpp] = np.cumsum(cur[k::bpp])
            out[y] = (lanes & 0xFF).astype(np.uint8)
        elif ft in (3, 4):                                     # Average / Paeth depend on the decoded left neighbour
            cur = out[y]
            ln = line.astype(np.int32); pr = prior.astype(np.int32)
            row = [0] * stride
            for x in range(stride):
                a = row[x - bpp] if x >= bpp else 0
                b = int(pr[x])
                if ft == 3:
                    v = int(ln[x]) + ((a + b) >> 1)
                else:
                    c = int(pr[x - bpp]) if x >= bpp else 0
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                    v = int(ln[x]) + pred
                row[x] = v & 0xFF
            cur[:] = np.asarray(row, dtype=np.uint8)
        else:
            raise PngError(f"bad filter type {ft}")
    return out[1:]


def read_png(data: bytes, channels: int) -> Tuple[np.ndarray, int, int]:
    """Decode to uint8[h, w, channels] with channels 3 or 4 (the two forms ``qoiconv.c:49-55`` asks stb_image for)."""
    if channels not in (3, 4):
        raise PngError("channels must be 3 or 4")
    ihdr = None; plte = None; trns = None; idat = []
    for typ, body in _chunks(data):
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"PLTE":
            plte = np.frombuffer(body, dtype=np.uint8).reshape(-1, 3)
        elif typ == b"tRNS":
            trns = body
        elif typ == b"IDAT":
            idat.append(body)
    if ihdr is None:
        raise PngError("no IHDR")
    w, h, depth, ctype, comp, flt, il = ihdr
    if ctype not in _SAMPLES or comp or flt or depth not in (1, 2, 4, 8, 16):
        raise PngError("unsupported PNG parameters")
    if il:
        raise PngError("interlaced PNGs are not supported")
    ns = _SAMPLES[ctype]
    stride = (w * ns * depth + 7) // 8
    bpp = max(1, ns * depth // 8)
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    if raw.size < h * (stride + 1):
        raise PngError("image data too short")
    rows = _unfilter(raw, h, stride, bpp)
    # samples -> uint16 array [h, w*ns] of raw sample values
    if depth == 8:
        s = rows.astype(np.uint16)
    elif depth == 16:
        s = (rows[:, 0::2].astype(np.uint16) << 8) | rows[:, 1::2]
    else:
        bits = np.unpackbits(rows, axis=1)[:, :w * ns * depth].reshape(h, w * ns, depth)
        s = np.zeros((h, w * ns), dtype=np.uint16)
        for k in range(depth):
            s = (s << 1) | bits[:, :, k]
    s = s.reshape(h, w, ns)
    maxv = (1 << depth) - 1
    alpha = np.full((h, w), 255, dtype=np.uint8)
    if ctype == 3:
        if plte is None:
            raise PngError("palette image without PLTE")
        idx = np.minimum(s[:, :, 0], len(plte) - 1)
        rgb = plte[idx]
        if trns is not None:
            ta = np.full(256, 255, dtype=np.uint8); ta[:len(trns)] = np.frombuffer(trns, dtype=np.uint8)
            alpha = ta[idx]
    else:
        def to8(v):
            return ((v >> 8) if depth == 16 else (v * 255 // maxv if depth < 8 else v)).astype(np.uint8)
        if ctype in (0, 4):
            g = to8(s[:, :, 0]); rgb = np.stack([g, g, g], axis=-1)
            if ctype == 4:
                alpha = to8(s[:, :, 1])
            elif trns is not None:
                key = struct.unpack(">H", trns[:2])[0]
                alpha = np.where(s[:, :, 0] == key, 0, 255).astype(np.uint8)
        else:
            rgb = to8(s[:, :, :3])
            if ctype == 6:
                alpha = to8(s[:, :, 3])
            elif trns is not None:
                key = struct.unpack(">HHH", trns[:6])
                hit = (s[:, :, 0] == key[0]) & (s[:, :, 1] == key[1]) & (s[:, :, 2] == key[2])
                alpha = np.where(hit, 0, 255).astype(np.uint8)
    out = np.empty((h, w, channels), dtype=np.uint8)
    out[:, :, :3] = rgb
    if channels == 4:
        out[:, :, 3] = alpha
    return out, w, h


def write_png(pixels: np.ndarray, level: int = 6) -> bytes:
    """uint8[h, w, 3|4] -> PNG bytes (colour type 2 or 6, 8 bit, filter 0 on every line; ``stbi_write_png`` role)."""
    px = np.ascontiguousarray(pixels, dtype=np.uint8)
    h, w, ch = px.shape
    if ch not in (3, 4):
        raise PngError("channels must be 3 or 4")
    raw = np.zeros((h, 1 + w * ch), dtype=np.uint8)
    raw[:, 1:] = px.reshape(h, w * ch)

    def chunk(typ: bytes, body: bytes) -> bytes:
        return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)

    return (_SIG + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if ch == 3 else 6, 0, 0, 0))
            + chunk(b"IDAT", zlib.compress(raw.tobytes(), level)) + chunk(b"IEND", b""))
