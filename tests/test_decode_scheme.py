"""CPU rehearsal of the GPU decoder's segment scheme (tests/host/decode_host.cpp).

The per-segment primitives of qoi_amd/csrc/qoi_decode_core.h are compiled with g++ and
driven in kernel-launch order; results must equal the reference decoder's (golden vectors)
for EVERY stream, at every segment size — including streams that defeat the slot
speculation and go through the restart loop.
"""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

import cases
from qoi_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("hostlib") / "libdecode_host.so")
    src = os.path.join(ROOT, "tests", "host", "decode_host.cpp")
    subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", out, src], check=True)
    lib = ctypes.CDLL(out)
    for fn in (lib.host_decode_pipeline_g, lib.host_decode_pipeline_fast, lib.host_decode_pipeline_rec):
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint32, ctypes.c_int, ctypes.c_uint32,
                       ctypes.c_uint32, ctypes.c_void_p, ctypes.POINTER(ctypes.c_longlong)]
    return lib


def run(lib, stream: bytes, och: int, B: int, grp: int = 64, fast=False):
    """fast: False = readable primitives, True = lean LUT-driven ones, "rec" = the chunk-record pipeline of the kernels"""
    w, h = struct.unpack(">II", stream[4:12])
    npx = w * h
    buf = np.frombuffer(stream + b"\0" * 8, dtype=np.uint8).copy()
    out = np.full(npx * och + 8, 0xAB, dtype=np.uint8)
    stats = (ctypes.c_longlong * 4)()
    fn = lib.host_decode_pipeline_rec if fast == "rec" else lib.host_decode_pipeline_fast if fast else lib.host_decode_pipeline_g
    fn(buf.ctypes.data, len(stream), npx, och, B, grp, out.ctypes.data, stats)
    return out[:npx * och], list(stats)


def test_scheme_matches_golden(host_lib, golden, encoded_streams):
    n = 0
    for c in cases.decode_cases(encoded_streams):
        if not bool(golden[f"dec/{c['name']}/ok"][0]):
            continue                      # rejections are host-side argument checks, not the pipeline
        desc = golden[f"dec/{c['name']}/desc"]
        och = c["channels"] if c["channels"] else int(desc[2])
        want = golden[f"dec/{c['name']}/pixels"]
        for B, grp in ((5, 3), (7, 64), (16, 2), (64, 5), (333, 64), (2048, 64), (4096, 64)):
            for fast in (False, True, "rec"):     # readable primitives, the lean LUT-driven ones, the record pipeline of the kernels
                got, stats = run(host_lib, c["stream"], och, B, grp, fast)
                assert np.array_equal(got, want), (c["name"], B, grp, fast, stats)
                assert stats[3] == 0, ("slot transfer from the record tail differs from the forward walk", c["name"], B)
        n += 1
    assert n > 100


def test_chunk_lut_matches_grammar(host_lib):
    """len_of / the 256-entry chunk table agree with chunk_len / chunk_pixels (qoi.h:547-575) for every tag byte."""
    assert host_lib.host_check_lut() == 0


def test_chunk_records_match_grammar(host_lib):
    """rec_of_chunk (the transcoder's record per chunk) carries exactly what crack() reads from the chunk bytes."""
    assert host_lib.host_check_records() == 0


def test_speculation_holds_on_encoder_streams(host_lib, port):
    """Encoder-made opaque content must verify in ONE round (no restart)."""
    for kind in ("photo", "noise", "constant"):
        w, h = 256, 192
        px = synth.frame_rgba(kind, w, h, 2)
        s = port.encode(px, w, h, 4)
        for fast in (False, True, "rec"):
            got, stats = run(host_lib, s, 4, 256, 64, fast)
            assert np.array_equal(got, px.reshape(-1))
            assert stats[0] == 1 and stats[1] == 0, (kind, fast, stats)


def test_uiflat_exact_even_if_restarts(host_lib, port):
    w, h = 400, 300
    px = synth.frame_rgba("uiflat", w, h, 1)
    s = port.encode(px, w, h, 4)
    for fast in (False, "rec"):
        got, stats = run(host_lib, s, 4, 64, 64, fast)
        assert np.array_equal(got, px.reshape(-1))


def test_refinement_passes_cut_the_rounds_of_flat_frames(host_lib, port, monkeypatch):
    """UI frames (alpha levels through the colour table): P3 + S3 repeated inside a refinement round, and appended to the first
    round of flat images, verify them in a few rounds; one pass per round crawls two segments per round through some stretches.
    Same schedule as the launcher (qoi_decode.hip launch_decode_round)."""
    w, h = 3840, 2160
    worst = {}
    for seed in (81, 73, 12):
        px = synth.frame_rgba("uiflat", w, h, seed)
        s = port.encode(px, w, h, 4)
        for name, env in (("default", {}), ("off", {"QOIMI_DEC_INNER": "1", "QOIMI_DEC_INNER1": "0"})):
            for k in ("QOIMI_DEC_INNER", "QOIMI_DEC_INNER1"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            got, stats = run(host_lib, s, 4, 512, 64, "rec")
            assert np.array_equal(got, px.reshape(-1)), (seed, name)
            worst[name] = max(worst.get(name, 0), stats[0])
    assert worst["default"] <= 5 and worst["off"] >= 2 * worst["default"], worst


def test_record_dense_segments(host_lib, port):
    """B + 1 records from a B-byte segment (cases.dense_record_streams): the record region of a segment must hold them."""
    n = 0
    for name, B, stream, w, h in cases.dense_record_streams():
        want, _ = port.decode(stream, 4)
        for fast in (True, "rec"):
            got, stats = run(host_lib, stream, 4, B, 64, fast)
            assert np.array_equal(got, want), (name, fast, stats)
        n += 1
    assert n == 21


def test_record_pairs(host_lib, port):
    """QOI_OP_RGB / QOI_OP_RGBA leave pairs of records on even record indices (cases.pair_streams; transcode_segment pads with null
    records as dec_transcode does): the record pipeline stays exact at every segment size, 3- and 4-channel output."""
    n = 0
    for name, stream, w, h in cases.pair_streams():
        if len(stream) > 250000:
            continue                                   # the long mixture is the GPU test's
        for och in (4, 3):
            want, _ = port.decode(stream, och)
            for B in (64, 128, 4096):
                got, stats = run(host_lib, stream, och, B, 64, "rec")
                assert np.array_equal(got, want), (name, och, B, stats)
                assert stats[3] == 0, ("slot transfer from the record tail differs from the forward walk", name, B)
        n += 1
    assert n == 5
