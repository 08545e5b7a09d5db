"""N4: png <> qoi conversion tool.  CPU part: the PNG reader/writer the tool brings along
(the reference uses stb_image, a third-party header).  GPU part: conversions through the library's qoi_write /
qoi_read, checked against the oracle."""
import os
import struct
import subprocess
import sys
import zlib

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import png_io  # noqa: E402

TOOL = os.path.join(ROOT, "tools", "qoiconv_mi355x.py")


def _chunk(typ, body):
    return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xFFFFFFFF)


def _filtered(rows: np.ndarray, bpp: int, types) -> bytes:
    """Apply PNG filters (spec 9.2) row by row with the given filter types - the inverse of what the reader does."""
    h, stride = rows.shape
    out = bytearray()
    prev = np.zeros(stride, dtype=np.int32)
    for y in range(h):
        cur = rows[y].astype(np.int32)
        ft = types[y % len(types)]
        a = np.concatenate([np.zeros(bpp, dtype=np.int32), cur[:-bpp]])
        c = np.concatenate([np.zeros(bpp, dtype=np.int32), prev[:-bpp]])
        b = prev
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = b
        elif ft == 3:
            pred = (a + b) >> 1
        else:
            p = a + b - c
            pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))
        out.append(ft)
        out += ((cur - pred) & 0xFF).astype(np.uint8).tobytes()
        prev = cur
    return bytes(out)


def _png(w, h, depth, ctype, rows, bpp, types=(0, 1, 2, 3, 4), extra=b""):
    return (b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, ctype, 0, 0, 0)) + extra
            + _chunk(b"IDAT", zlib.compress(_filtered(rows, bpp, types))) + _chunk(b"IEND", b""))


def test_png_all_filters_rgba_and_python_fallback():
    rng = np.random.default_rng(1)
    w, h = 37, 23
    px = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    data = _png(w, h, 8, 6, px.reshape(h, w * 4), 4)
    assert png_io.png_info(data) == (w, h, 4)
    got, gw, gh = png_io.read_png(data, 4)
    assert (gw, gh) == (w, h) and np.array_equal(got, px)
    got3, _, _ = png_io.read_png(data, 3)
    assert np.array_equal(got3, px[:, :, :3])
    # the pure-numpy reconstruction agrees with the native one
    raw = np.frombuffer(zlib.decompress(b"".join(b for t, b in png_io._chunks(data) if t == b"IDAT")), dtype=np.uint8)
    assert np.array_equal(png_io._unfilter_py(raw, h, w * 4, 4), px.reshape(h, w * 4))


def test_png_colour_types_and_depths():
    rng = np.random.default_rng(2)
    w, h = 19, 11
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    d = _png(w, h, 8, 2, rgb.reshape(h, w * 3), 3)
    assert png_io.png_info(d)[2] == 3
    assert np.array_equal(png_io.read_png(d, 3)[0], rgb)
    assert np.array_equal(png_io.read_png(d, 4)[0][:, :, 3], np.full((h, w), 255))
    g = rng.integers(0, 256, (h, w), dtype=np.uint8)                      # grey -> RGBA, like stbi_load(..., 4)
    d = _png(w, h, 8, 0, g, 1)
    assert png_io.png_info(d)[2] == 1
    o = png_io.read_png(d, 4)[0]
    assert np.array_equal(o[:, :, 0], g) and np.array_equal(o[:, :, 2], g) and (o[:, :, 3] == 255).all()
    ga = rng.integers(0, 256, (h, w, 2), dtype=np.uint8)                  # grey + alpha
    o = png_io.read_png(_png(w, h, 8, 4, ga.reshape(h, w * 2), 2), 4)[0]
    assert np.array_equal(o[:, :, 1], ga[:, :, 0]) and np.array_equal(o[:, :, 3], ga[:, :, 1])
    pal = rng.integers(0, 256, (5, 3), dtype=np.uint8)                    # palette + tRNS, 4 bits per index
    idx = rng.integers(0, 5, (h, w), dtype=np.uint8)
    packed = np.zeros((h, (w + 1) // 2), dtype=np.uint8)
    for x in range(w):
        packed[:, x // 2] |= idx[:, x] << (4 if x % 2 == 0 else 0)
    extra = _chunk(b"PLTE", pal.tobytes()) + _chunk(b"tRNS", bytes([0, 128]))
    d = _png(w, h, 4, 3, packed, 1, extra=extra)
    assert png_io.png_info(d)[2] == 4
    o = png_io.read_png(d, 4)[0]
    assert np.array_equal(o[:, :, :3], pal[idx])
    assert np.array_equal(o[:, :, 3], np.array([0, 128, 255, 255, 255], dtype=np.uint8)[idx])
    s16 = rng.integers(0, 65536, (h, w, 3), dtype=np.uint16)              # 16-bit samples keep the high byte
    be = s16.astype(">u2").tobytes()
    o = png_io.read_png(_png(w, h, 16, 2, np.frombuffer(be, dtype=np.uint8).reshape(h, w * 6), 6), 3)[0]
    assert np.array_equal(o, (s16 >> 8).astype(np.uint8))


def test_png_writer_round_trip_and_errors():
    rng = np.random.default_rng(3)
    for ch in (3, 4):
        px = rng.integers(0, 256, (13, 29, ch), dtype=np.uint8)
        blob = png_io.write_png(px)
        assert png_io.png_info(blob) == (29, 13, ch)
        assert np.array_equal(png_io.read_png(blob, ch)[0], px)
    good = png_io.write_png(np.zeros((2, 2, 3), dtype=np.uint8))
    with pytest.raises(png_io.PngError):
        png_io.read_png(b"not a png", 4)
    bad = bytearray(good); bad[20] ^= 1                                   # IHDR CRC no longer matches
    with pytest.raises(png_io.PngError):
        png_io.png_info(bytes(bad))
    with pytest.raises(png_io.PngError):
        png_io.read_png(good[:-20], 3)                                    # truncated


def test_usage_message_and_exit_code():
    r = subprocess.run([sys.executable, TOOL], capture_output=True, text=True)
    assert r.returncode == 1 and r.stdout.startswith("Usage: qoiconv <infile> <outfile>")   # qoiconv.c:34-41


@pytest.mark.gpu
def test_png_to_qoi_to_png(tmp_path, ref, port):
    oracle = ref or port
    from qoi_amd import synth
    for ch, name in ((4, "a"), (3, "b")):
        px = synth.frame_rgba("photo" if ch == 3 else "uiflat", 200, 120, 5)
        src = px if ch == 4 else np.ascontiguousarray(px[:, :, :3])
        png1, qoi, png2 = (str(tmp_path / f"{name}{e}") for e in ("1.png", ".qoi", "2.png"))
        open(png1, "wb").write(png_io.write_png(src))
        assert subprocess.run([sys.executable, TOOL, png1, qoi]).returncode == 0
        blob = open(qoi, "rb").read()
        assert blob == oracle.encode(src, 200, 120, ch, 0)                 # what the reference's qoiconv would write
        assert subprocess.run([sys.executable, TOOL, qoi, png2]).returncode == 0
        assert np.array_equal(png_io.read_png(open(png2, "rb").read(), ch)[0], src)
    assert subprocess.run([sys.executable, TOOL, str(tmp_path / "missing.png"), str(tmp_path / "x.qoi")]).returncode == 1
