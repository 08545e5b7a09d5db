"""Deterministic input cases for the parity suites.

The reference holds NO golden vectors or unit tests for this path (SURVEY.md §4,
§8c); its only in-tree check is the encode->decode->memcmp round trip of
qoibench.c:408-417.  These cases are therefore ours; the EXPECTED outputs come
from the unmodified reference (tests/golden/make_golden.py -> tests/golden/*.npz).
The list follows SURVEY.md Appendix B (bit-exactness checklist).
"""
from __future__ import annotations

import struct
from typing import Dict, List

import numpy as np

from qoi_amd import synth

MAGIC = b"qoif"
END = b"\0\0\0\0\0\0\0\x01"


def header(w: int, h: int, ch: int = 4, cs: int = 0, magic: bytes = MAGIC) -> bytes:
    return magic + struct.pack(">II", w, h) + bytes([ch, cs])


def _rng(tag: str) -> np.random.Generator:
    return np.random.default_rng(abs(hash_str(tag)) % (2 ** 32))


def hash_str(s: str) -> int:
    h = 2166136261
    for c in s.encode():
        h = ((h ^ c) * 16777619) & 0xFFFFFFFF
    return h


def _px(*rgba) -> np.ndarray:
    return np.array(rgba, dtype=np.uint8)


def encode_cases() -> List[Dict]:
    """Each case: name, w, h, ch, cs, pixels (uint8, len w*h*ch) or pixels=None for arg checks."""
    cases: List[Dict] = []

    def add(name, w, h, ch, pixels, cs=0):
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8).reshape(-1)
        assert pixels.size == w * h * ch, (name, pixels.size, w, h, ch)
        cases.append(dict(name=name, w=w, h=h, ch=ch, cs=cs, pixels=pixels))

    # synthetic content classes, ragged sizes, both channel counts
    for kind in synth.KINDS:
        for (w, h) in [(64, 48), (37, 23), (257, 9)]:
            add(f"synth_{kind}_{w}x{h}_c4", w, h, 4, synth.frame_rgba(kind, w, h, 3))
            add(f"synth_{kind}_{w}x{h}_c3", w, h, 3, synth.frame_rgb(kind, w, h, 3))
    add("photo_200x150_linear", 200, 150, 4, synth.frame_rgba("photo", 200, 150, 5), cs=1)

    # Appendix B item 12: 1x1 images
    add("1x1_initial_prev", 1, 1, 4, _px(0, 0, 0, 255))          # -> single RUN chunk 0xC0
    add("1x1_zero", 1, 1, 4, _px(0, 0, 0, 0))                    # -> INDEX hit on zeroed slot 0
    add("1x1_other", 1, 1, 4, _px(9, 200, 3, 77))
    add("1x1_c3", 1, 1, 3, _px(1, 2, 3))
    add("1x1_c3_black", 1, 1, 3, _px(0, 0, 0))                   # equals initial prev with a=255

    # run lengths around the 62 cap (qoi.h:417) incl. runs against the initial prev
    for n in (2, 61, 62, 63, 64, 123, 124, 125, 126, 200):
        add(f"run_initial_{n}", n, 1, 4, np.tile(_px(0, 0, 0, 255), n))
        body = np.concatenate([_px(10, 20, 30, 255), np.tile(_px(10, 20, 30, 255), n), _px(1, 1, 1, 255)])
        add(f"run_mid_{n}", n + 2, 1, 4, body)
        add(f"run_tail_{n}", n + 1, 1, 4, np.concatenate([_px(5, 6, 7, 8), np.tile(_px(5, 6, 7, 8), n)]))
    # initial run of {0,0,0,255} then the same colour again later: the encoder did NOT index it
    add("initial_run_then_same_colour", 6, 1, 4,
        np.concatenate([np.tile(_px(0, 0, 0, 255), 3), _px(1, 2, 3, 255), _px(0, 0, 0, 255), _px(0, 0, 0, 255)]))

    # hash collisions: colours sharing a slot evict each other (hash = 3r+5g+7b+11a & 63)
    a = _px(0, 0, 0, 255)      # slot 53
    b = _px(64, 0, 0, 255)     # 3*64 = 192 = 0 mod 64 -> same slot 53
    c = _px(0, 64, 0, 255)     # same slot
    add("collide_alternate", 12, 1, 4, np.concatenate([b, c, b, c, a, b, a, c, c, b, a, a]))
    # all 64 slots then revisit in reverse
    cols = np.array([[i, 0, 0, 255] for i in range(0, 64)], dtype=np.uint8)  # slots 3i+53 mod 64: all distinct
    add("all_slots_revisit", 128, 1, 4, np.concatenate([cols, cols[::-1]]))

    # alpha transitions (RGBA op) and 8-bit wrap-around of DIFF / LUMA deltas
    add("alpha_steps", 8, 1, 4, np.array([[10, 10, 10, 255], [10, 10, 10, 254], [11, 10, 10, 254], [11, 10, 10, 0],
                                          [0, 0, 0, 0], [255, 255, 255, 0], [0, 1, 255, 0], [0, 1, 255, 255]], dtype=np.uint8))
    ramp = np.array([[(250 + i) & 255, (253 + 2 * i) & 255, (5 - i) & 255, 255] for i in range(16)], dtype=np.uint8)
    add("wrap_diff", 16, 1, 4, ramp)
    luma = np.array([[(i * 13) & 255, (i * 17) & 255, (i * 11) & 255, 255] for i in range(40)], dtype=np.uint8)
    add("wrap_luma", 40, 1, 4, luma)
    # boundaries of the DIFF (-2..1) and LUMA (-32..31, -8..7) windows
    edge = [[100, 100, 100, 255]]
    for (dr, dg, db) in [(-2, -2, -2), (1, 1, 1), (-3, 0, 0), (2, 0, 0), (0, -32, 0), (0, 31, 0), (0, -33, 0), (0, 32, 0),
                         (-8 + 5, 5, 7 + 5), (-9 + 5, 5, 5), (8 + 5, 5, 5), (5, 5, -9 + 5), (5, 5, 8 + 5), (0, 0, 0)]:
        p = edge[-1]
        edge.append([(p[0] + dr) & 255, (p[1] + dg) & 255, (p[2] + db) & 255, 255])
    add("op_window_edges", len(edge), 1, 4, np.array(edge, dtype=np.uint8))

    # small palettes (INDEX heavy) and random noise at odd sizes
    for k in (2, 8, 24, 80):
        r = _rng(f"pal{k}")
        pal = r.integers(0, 256, size=(k, 4), dtype=np.uint8)
        idx = r.integers(0, k, size=61 * 17)
        add(f"palette_{k}_c4", 61, 17, 4, pal[idx])
        add(f"palette_{k}_c3", 61, 17, 3, pal[idx][:, :3])
    r = _rng("sticky")
    v = r.integers(0, 256, size=(3000, 4), dtype=np.uint8)
    keep = r.random(3000) < 0.7
    for i in range(1, 3000):
        if keep[i]:
            v[i] = v[i - 1]
    add("sticky_runs", 100, 30, 4, v)
    return cases


def encode_arg_cases() -> List[Dict]:
    """Argument validation of qoi.h:364-372; pixels are never read, expected result NULL."""
    return [
        dict(name="w0", w=0, h=4, ch=4, cs=0),
        dict(name="h0", w=4, h=0, ch=4, cs=0),
        dict(name="ch2", w=4, h=4, ch=2, cs=0),
        dict(name="ch5", w=4, h=4, ch=5, cs=0),
        dict(name="cs2", w=4, h=4, ch=4, cs=2),
        dict(name="too_many_px", w=20000, h=20000, ch=4, cs=0),
        dict(name="cap_edge", w=16384, h=24414, ch=4, cs=0),   # 400000000/16384 = 24414 -> rejected
    ]


def _chunks(*parts) -> bytes:
    return b"".join(bytes(p) if not isinstance(p, bytes) else p for p in parts)


def decode_cases(encoded: Dict[str, bytes]) -> List[Dict]:
    """Each case: name, stream (bytes), channels (0/3/4), optional size override.

    ``encoded`` maps encode-case names to reference-encoded streams (used as bases
    for truncation / mutation).
    """
    cases: List[Dict] = []

    def add(name, stream, channels=4, size=None):
        cases.append(dict(name=name, stream=bytes(stream), channels=channels, size=size))

    # well-formed streams, all three channel arguments (Appendix B item 7)
    for nm in ("synth_photo_64x48_c4", "synth_photo_37x23_c3", "synth_noise_37x23_c4", "synth_uiflat_257x9_c4",
               "synth_constant_64x48_c4", "palette_24_c4", "palette_8_c3", "alpha_steps", "sticky_runs",
               "collide_alternate", "all_slots_revisit", "photo_200x150_linear"):
        for chn in (0, 3, 4):
            add(f"wf_{nm}_ch{chn}", encoded[nm], chn)

    base = encoded["synth_photo_64x48_c4"]
    # item 1: truncation at every offset near the front and at a spread of later offsets
    for cut in list(range(22, 60)) + [100, 333, len(base) // 2, len(base) - 9, len(base) - 1]:
        if cut < len(base):
            add(f"trunc_{cut}", base[:cut], 4)
    # item 2: no chunks at all
    add("size22", header(5, 3) + END, 4)
    add("size22_c3", header(5, 3, 3) + END, 0)
    # item 3: trailer content is not inspected
    add("garbage_trailer", base[:-8] + b"\xde\xad\xbe\xef\x01\x02\x03\x04", 4)
    # item 4: multi-byte chunk starting inside the chunk region reads into the trailer
    add("rgba_into_trailer", header(2, 1) + b"\xff\x01\x02" + b"\x03\x04\x05\x06\x07\x08\x09\x0a", 4)
    add("luma_into_trailer", header(3, 1) + b"\x40\xa0" + b"\x9c\x00\x00\x00\x00\x00\x00\x01", 4)
    # item 5: INDEX of a never-written slot
    add("index_unwritten", header(4, 1) + bytes([0x05, 0x00, 0x3f, 0x05]) + END, 4)
    # item 6: table is updated after RUN chunks too
    add("run_updates_table", header(5, 1) + bytes([0xC1, 0xFE, 5, 6, 7, 0x35, 0x35]) + END, 4)
    # item 7: RGBA op inside a 3-channel file, all output channel counts
    s = header(3, 1, 3) + bytes([0xFF, 1, 2, 3, 4, 0x40 | 0x2A, 0xFE, 9, 9, 9]) + END
    for chn in (0, 3, 4):
        add(f"rgba_in_c3_ch{chn}", s, chn)
    # item 8: over-long run clipped, extra chunks ignored
    add("run_clipped", header(3, 1) + bytes([0xFD, 0xFE, 1, 2, 3]) + END, 4)
    add("extra_chunks", header(2, 1) + bytes([0xFE, 1, 2, 3, 0x6A, 0xFE, 7, 7, 7, 0x00, 0xC5]) + END, 4)
    # item 9: wrap-around
    add("wrap", header(3, 1) + bytes([0x40, 0x80, 0x00, 0x7F]) + END, 4)
    # tags 0xFE/0xFF take precedence over the 2-bit RUN tag; payload bytes that look like tags
    add("payload_looks_like_tags", header(6, 1) + bytes([0xFE, 0xFF, 0xFE, 0xC0, 0xFF, 0xC0, 0xFE, 0xFF, 0x00, 0xBF, 0xFF, 0x3F, 0xFD]) + END, 4)
    # a long INDEX chain through different slots (value-dependent dependencies)
    chain = bytearray()
    for i in range(64):
        chain += bytes([0xFE, (i * 37) & 255, (i * 11) & 255, (i * 5) & 255])
    for i in range(200):
        chain += bytes([(i * 29 + 7) & 63])
        if i % 7 == 0:
            chain += bytes([0x40 | ((i * 13) & 63)])
        if i % 11 == 0:
            chain += bytes([0x80 | (i & 63), (i * 3) & 255])
    add("index_chain", header(33, 10) + bytes(chain) + END, 4)

    # item 10: rejections -> NULL
    add("rej_size21", (header(1, 1) + END)[:21], 4)
    add("rej_size0", b"", 4, size=0)
    add("rej_channels1", base, 1)
    add("rej_channels5", base, 5)
    add("rej_magic", header(2, 2, magic=b"qoig") + bytes([0xC3]) + END, 4)
    add("rej_hdr_ch2", header(2, 2, 2) + bytes([0xC3]) + END, 4)
    add("rej_hdr_ch5", header(2, 2, 5) + bytes([0xC3]) + END, 4)
    add("rej_hdr_cs2", header(2, 2, 4, 2) + bytes([0xC3]) + END, 4)
    add("rej_w0", header(0, 2) + bytes([0xC3]) + END, 4)
    add("rej_h0", header(2, 0) + bytes([0xC3]) + END, 4)
    add("rej_too_many_px", header(20000, 20000) + bytes([0xC3]) + END, 4)

    # seeded fuzz: random chunk soup behind valid headers (qoifuzz.c:20-32 style, but with expected outputs)
    r = _rng("fuzz")
    for i in range(60):
        w = int(r.integers(1, 40)); h = int(r.integers(1, 12))
        n = int(r.integers(0, 3 * w * h + 8))
        kind = i % 3
        if kind == 0:
            body = r.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        elif kind == 1:   # biased towards INDEX / DIFF / short runs so INDEX chains are long
            body = r.choice(np.array([0x00, 0x01, 0x15, 0x2A, 0x3F, 0x55, 0x6A, 0x7F, 0x95, 0xC0, 0xC1, 0xFE, 0xFF], dtype=np.uint8),
                            size=n).tobytes()
        else:             # mutate a valid stream
            src = bytearray(encoded["palette_24_c4"][14:14 + n])
            for _ in range(max(1, len(src) // 16)):
                if src:
                    src[int(r.integers(0, len(src)))] = int(r.integers(0, 256))
            body = bytes(src)
        add(f"fuzz_{i}", header(w, h, 3 + (i & 1)) + body + END, (0, 3, 4)[i % 3])
    return cases


def dense_record_streams():
    """Streams whose segments are as dense in chunk records as the format allows: one-byte chunks up to the last byte of a
    B-byte decode segment and a QOI_OP_RGBA that STARTS on that last byte (its five bytes reach into the next segment, both of
    its records belong to this one) - B + 1 records from B bytes.  Round 2's first record pipeline reserved B and lost pixels on
    1 frame in 60 of the `uiflat` content at 512-byte segments.  Yields (name, B, stream, width, height)."""
    import struct
    for B in (64, 128, 256, 512, 1024, 2048, 4096):
        for lead in (0, 1, 3):                      # the dense segment as segment 0 / after a few other chunks
            body = bytearray()
            body += bytes([0xFE, 10, 20, 30])       # an RGB first, so runs and INDEX have something to repeat
            while (len(body) + lead) % B != B - 1 or len(body) < 2 * B:
                k = len(body) % 7
                body.append(0xC0 | (k % 5) if k < 4 else (0x40 | (k * 9 & 0x3F)) if k < 6 else (k * 5 & 0x3F))
            body = bytearray(bytes([0xC1] * lead)) + body
            assert len(body) % B == B - 1
            body += bytes([0xFF, 1, 2, 3, 0x80])    # starts on the last byte of its segment
            body += bytes([0x6A, 0xC3, 0x15, 0xFF, 9, 8, 7, 6, 0x3F, 0xC0]) * 3
            npx = 0
            i = 0
            while i < len(body):
                b = body[i]
                n = 4 if b == 0xFE else 5 if b == 0xFF else 2 if (b >> 6) == 2 else 1
                npx += (b & 0x3F) + 1 if (b >> 6) == 3 and b < 0xFE else 1
                i += n
            w = 64
            h = (npx + w - 1) // w + 1              # a few pixels past the last chunk: filled with the last pixel (qoi.h:544)
            stream = b"qoif" + struct.pack(">II", w, h) + bytes([4, 0]) + bytes(body) + bytes([0, 0, 0, 0, 0, 0, 0, 1])
            yield f"dense_B{B}_lead{lead}", B, stream, w, h


def pair_streams():
    """Streams dense in QOI_OP_RGB / QOI_OP_RGBA, the chunks the transcoder lays out as PAIRS of records on even record
    indices (dec_transcode): all of them such chunks (every block of every lane takes the pair path of P3 / P4), nearly all
    (blocks fall back to the general steps at random, pairs are pushed to the next even index by one-byte chunks), half.
    Yields (name, stream, width, height)."""
    import struct
    for seed, p_hi, p_rgba, n_chunks in ((1, 1.0, 1.0, 40000), (2, 1.0, 0.5, 40000), (3, 1.0, 0.0, 40000), (4, 0.97, 0.7, 40000),
                                         (5, 0.5, 0.5, 60000), (6, 0.9, 0.2, 300000)):
        rng = np.random.default_rng(seed)
        body = bytearray()
        npx = 0
        kinds = rng.random(n_chunks)
        sub = rng.random(n_chunks)
        val = rng.integers(0, 256, size=(n_chunks, 4))
        for i in range(n_chunks):
            if kinds[i] < p_hi:
                if sub[i] < p_rgba:
                    body += bytes([0xFF, val[i, 0], val[i, 1], val[i, 2], val[i, 3]])
                else:
                    body += bytes([0xFE, val[i, 0], val[i, 1], val[i, 2]])
                npx += 1
            elif sub[i] < 0.3:
                body.append(val[i, 0] & 0x3F); npx += 1                        # INDEX
            elif sub[i] < 0.6:
                body.append(0x40 | (val[i, 0] & 0x3F)); npx += 1                 # DIFF
            elif sub[i] < 0.8:
                body += bytes([0x80 | (val[i, 0] & 0x3F), val[i, 1]]); npx += 1  # LUMA
            else:
                r = val[i, 0] % 62
                body.append(0xC0 | r); npx += r + 1                              # RUN
        w = 97
        h = (npx + w - 1) // w + 1
        yield (f"pairs_seed{seed}", b"qoif" + struct.pack(">II", w, h) + bytes([4, 0]) + bytes(body) + bytes([0, 0, 0, 0, 0, 0, 0, 1]), w, h)


def flat_run_streams():
    """Streams of FLAT images (less than a byte per eight pixels: what the decoder gives run descriptors, dec_segments_rec<OCH, true>
    + dec_expand_runs): long runs cut every 62 pixels (qoi.h:417) that begin at every alignment, end in every tail length, follow
    each other with one literal chunk between them or none, open the image, run over its end (clipped, Appendix B item 8) or stop
    short of it (the last pixel repeats, qoi.h:544); alpha levels that change through QOI_OP_RGBA and come back through QOI_OP_INDEX.
    Yields (name, stream, width, height)."""
    import struct
    for seed, n_items, variant in ((1, 300, "exact"), (2, 300, "clipped"), (3, 300, "short"), (4, 3000, "exact"), (5, 3000, "clipped"),
                                   (6, 40, "exact"), (7, 20000, "exact"), (8, 3000, "short")):
        rng = np.random.default_rng(1000 + seed)
        body = bytearray()
        npx = 0
        if seed % 2 == 0:                                           # the image opens with a run of the start value (qoi.h:396-399)
            body += bytes([0xFD] * int(rng.integers(1, 6))) + bytes([0xC0 | int(rng.integers(0, 62))])
        pal = rng.integers(0, 256, size=(12, 4))
        pal[:, 3] = rng.choice([255, 255, 128, 0], size=12)
        for it in range(n_items):
            k = rng.random()
            c = pal[int(rng.integers(0, 12))]
            if k < 0.35:
                body += bytes([0xFF, c[0], c[1], c[2], c[3]])       # QOI_OP_RGBA (alpha level)
            elif k < 0.55:
                body += bytes([0xFE, c[0], c[1], c[2]])             # QOI_OP_RGB
            elif k < 0.75:
                body.append(int(rng.integers(0, 64)))               # QOI_OP_INDEX (often a colour seen before)
            elif k < 0.85:
                body.append(0x40 | int(rng.integers(0, 64)))        # QOI_OP_DIFF
            elif k < 0.9:
                body += bytes([0x80 | int(rng.integers(0, 64)), int(rng.integers(0, 256))])   # QOI_OP_LUMA
            # else: no chunk - the next run follows the previous one directly
            # a run of 1 .. ~2500 pixels: full QOI_OP_RUN chunks of 62 and a rest
            total = int(rng.choice([1, 2, 3, 5, 11, 12, 13, 14, 15, 16, 61, 62, 63, 64, 100, 124, 125, 300, 1000, 2500])) + int(rng.integers(0, 4))
            full, rest = divmod(total, 62)
            body += bytes([0xFD] * full)
            if rest:
                body.append(0xC0 | (rest - 1))
        # pixels the chunks produce
        i = 0
        while i < len(body):
            b = body[i]
            i += 4 if b == 0xFE else 5 if b == 0xFF else 2 if (b >> 6) == 2 else 1
            npx += (b & 0x3F) + 1 if (b >> 6) == 3 and b < 0xFE else 1
        w = 251 if seed != 6 else 17
        if variant == "exact":
            h = max(1, npx // w)                                    # (the last partial row of chunks is cut off: clipped as well)
        elif variant == "clipped":
            h = max(1, (npx - int(rng.integers(1, 2000))) // w)
        else:
            h = (npx + w - 1) // w + int(rng.integers(1, 40))
        while w * h < 8 * (len(body) + 14):                         # keep it flat (dec_image_is_flat): more rows of the last pixel
            h += 7
        yield (f"flat_seed{seed}_{variant}", b"qoif" + struct.pack(">II", w, h) + bytes([4, 0]) + bytes(body) + bytes([0, 0, 0, 0, 0, 0, 0, 1]), w, h)
