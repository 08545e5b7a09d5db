"""Pins the CPU oracle (oracle/qoi_oracle.c) to the reference.

(1) against the committed golden vectors produced by the unmodified reference
    (tests/golden/make_golden.py), and
(2) live against oracle/_ref/libqoiref.so on seeded random inputs when that
    build is present.
CPU only.
"""
import zlib

import numpy as np
import pytest

import cases
from oracle import oracle_py as O
from qoi_amd import synth


def test_desc_abi():
    import ctypes
    assert ctypes.sizeof(O.QoiDesc) == 12
    assert O.QoiDesc.width.offset == 0 and O.QoiDesc.height.offset == 4
    assert O.QoiDesc.channels.offset == 8 and O.QoiDesc.colorspace.offset == 9


def test_cases_match_golden_inputs(golden):
    for c in cases.encode_cases():
        assert zlib.crc32(c["pixels"].tobytes()) == int(golden[f"enc/{c['name']}/crc_in"][0]), c["name"]


def test_port_encode_golden(port, golden):
    for c in cases.encode_cases():
        s = port.encode(c["pixels"], c["w"], c["h"], c["ch"], c["cs"])
        assert s == golden[f"enc/{c['name']}/stream"].tobytes(), c["name"]


def test_port_encode_rejects(port):
    dummy = np.zeros(16, dtype=np.uint8)
    for c in cases.encode_arg_cases():
        p, _ = port.encode_raw(dummy.ctypes.data, O.QoiDesc(c["w"], c["h"], c["ch"], c["cs"]))
        assert p == 0, c["name"]


def test_port_decode_golden(port, golden, encoded_streams):
    n = 0
    for c in cases.decode_cases(encoded_streams):
        assert c["stream"] == golden[f"dec/{c['name']}/stream"].tobytes(), c["name"]
        px, d = port.decode(c["stream"], c["channels"], c["size"])
        ok = bool(golden[f"dec/{c['name']}/ok"][0])
        assert (px is not None) == ok, c["name"]
        if len(c["stream"]) >= 22 and c["channels"] in (0, 3, 4):
            assert [d.width, d.height, d.channels, d.colorspace] == list(golden[f"dec/{c['name']}/desc"]), c["name"]
        if ok:
            assert np.array_equal(px, golden[f"dec/{c['name']}/pixels"]), c["name"]
        n += 1
    assert n > 150


def test_round_trip_property(port):
    # the reference's own (only) check: qoibench.c:408-417
    for kind in synth.KINDS:
        px = synth.frame_rgba(kind, 320, 200, 7)
        s = port.encode(px, 320, 200, 4)
        back, d = port.decode(s, 4)
        assert (d.width, d.height, d.channels) == (320, 200, 4)
        assert np.array_equal(back, px.reshape(-1))


def test_port_vs_reference_live(port, ref):
    if ref is None:
        pytest.skip("oracle/_ref/libqoiref.so not built (no /root/reference here)")
    rng = np.random.default_rng(2024)
    for it in range(300):
        w = int(rng.integers(1, 70)); h = int(rng.integers(1, 40)); ch = 3 + int(rng.integers(0, 2))
        mode = it % 5
        if mode == 0:
            px = rng.integers(0, 256, size=(w * h, ch), dtype=np.uint8)
        elif mode == 1:
            pal = rng.integers(0, 256, size=(int(rng.integers(1, 40)), ch), dtype=np.uint8)
            px = pal[rng.integers(0, len(pal), size=w * h)]
        elif mode == 2:
            px = np.cumsum(rng.integers(-3, 4, size=(w * h, ch)), axis=0).astype(np.uint8)
        elif mode == 3:
            px = np.repeat(rng.integers(0, 256, size=(w * h, ch), dtype=np.uint8), 1, axis=0)
            stick = rng.random(w * h) < 0.8
            for i in range(1, w * h):
                if stick[i]:
                    px[i] = px[i - 1]
        else:
            px = synth.frame_rgba(synth.KINDS[it % 4], w, h, it)[..., :ch].reshape(-1, ch)
        a = port.encode(px, w, h, ch); b = ref.encode(px, w, h, ch)
        assert a == b, (it, w, h, ch)
        for chn in (0, 3, 4):
            pa, da = port.decode(a, chn); pb, db = ref.decode(a, chn)
            assert np.array_equal(pa, pb)
        # corrupt the stream: both decoders must agree byte-for-byte
        s = bytearray(a)
        for _ in range(4):
            s[int(rng.integers(14, len(s)))] = int(rng.integers(0, 256))
        cut = int(rng.integers(22, len(s) + 1))
        pa, da = port.decode(bytes(s[:cut]), 4); pb, db = ref.decode(bytes(s[:cut]), 4)
        assert (pa is None) == (pb is None)
        if pa is not None:
            assert np.array_equal(pa, pb)


def test_port_vs_reference_on_the_fuzzers_inputs(port, ref):
    """The generators of the GPU fuzzers (tests/fuzz_encode.py: patchwork images with runs across the 62 cap, colours in one hash
    slot, alpha steps, all-zero and start-value pixels; tests/fuzz_decode_batch.py: hostile valid-grammar chunk soups, too short and
    too long for their image) against the C restatement on the CPU: its streams and pixels must equal the unmodified reference's -
    the restatement is the checker on boxes without oracle/_ref, and these are the inputs that found a deviation in the GPU encoder."""
    if ref is None:
        pytest.skip("oracle/_ref not built")
    import fuzz_decode_batch
    import fuzz_encode
    rng = np.random.default_rng(2024)
    for _ in range(120):
        w, h = fuzz_encode.random_shape(rng, 400_000)
        ch = int(rng.choice([3, 4]))
        img = np.ascontiguousarray(fuzz_encode.random_image(rng, w, h)[:, :, :ch])
        a, b = port.encode(img, w, h, ch), ref.encode(img, w, h, ch)
        assert a == b, (w, h, ch)
        pa, _ = port.decode(a, 0)
        assert np.array_equal(pa, img.reshape(-1))
    for _ in range(120):
        w, h, ch = int(rng.integers(1, 500)), int(rng.integers(1, 300)), int(rng.choice([3, 4]))
        s = fuzz_decode_batch.random_stream(rng, w, h, ch, ref)
        for och in (0, 3, 4):
            pa, da = port.decode(s, och)
            pb, db = ref.decode(s, och)
            assert (pa is None) == (pb is None) and (da.width, da.height, da.channels, da.colorspace) == (db.width, db.height, db.channels, db.colorspace)
            assert pa is None or np.array_equal(pa, pb), (w, h, ch, och, len(s))
