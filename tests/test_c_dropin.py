"""The link-time swap of INTEGRATION.md section 1, for real: tests/c_dropin/dropin_main.c is a C caller of the reference's
public API compiled against the reference's header WITHOUT QOI_IMPLEMENTATION and linked with libqoi_mi355x.so
(tests/c_dropin/Makefile, run by __graft_entry__.build()).  CPU part: it builds and links against that header.  GPU part:
qoibench.c:408-417's round trip, qoiconv.c's qoi_write / qoi_read pattern and qoifuzz.c:20-32's decode call run through it
and are compared with the oracle."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_bin", "dropin")


def _build():
    subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "c_dropin")], check=True, capture_output=True)


def test_c_caller_links_against_reference_header():
    _build()
    assert os.path.exists(BIN)
    which = subprocess.run([BIN, "header"], capture_output=True, text=True, check=True).stdout.strip()
    if os.path.exists("/root/reference/qoi.h"):
        assert which == "reference qoi.h"
    # the library's four drop-in symbols are what the C translation unit left undefined
    und = subprocess.run(["nm", "-u", BIN], capture_output=True, text=True, check=True).stdout
    for sym in ("qoi_encode", "qoi_decode", "qoi_write", "qoi_read"):
        assert sym in und, sym


@pytest.mark.gpu
def test_c_caller_round_trips():
    if not os.path.exists(BIN):
        _build()
    assert os.path.exists(BIN), "tests/_bin/dropin not built (python -c 'import __graft_entry__ as g; g.build()')"
    for w, h, ch in ((64, 64, 4), (640, 481, 3), (1920, 1080, 4)):
        r = subprocess.run([BIN, "roundtrip", str(w), str(h), str(ch)], capture_output=True, text=True)
        assert r.returncode == 0 and r.stdout.startswith("ok "), (w, h, ch, r.stdout, r.stderr)


@pytest.mark.gpu
def test_c_caller_fuzz_inputs_match_oracle(tmp_path, golden, encoded_streams, ref, port):
    """qoifuzz.c's input convention (4-byte channels prefix); result compared with the golden vectors of the reference."""
    if not os.path.exists(BIN):
        _build()
    picked = []
    for c in cases.decode_cases(encoded_streams):
        if c["size"] is not None and c["size"] != len(c["stream"]):
            continue
        f = tmp_path / f"case{len(picked)}.bin"
        f.write_bytes(struct.pack("<i", c["channels"]) + c["stream"])
        picked.append((c, f))
    assert len(picked) > 100
    # ONE process for all of them (one line of output per file), plus a few as processes of their own as qoifuzz runs them
    outs = subprocess.run([BIN, "fuzz"] + [str(f) for _, f in picked], capture_output=True, text=True, check=True).stdout.strip().split("\n")
    assert len(outs) == len(picked)
    for (c, f) in picked[::40]:
        assert subprocess.run([BIN, "fuzz", str(f)], capture_output=True, text=True, check=True).stdout.strip() == outs[picked.index((c, f))]
    for (c, _), out in zip(picked, outs):
        ok = bool(golden[f"dec/{c['name']}/ok"][0])
        if not ok:
            assert out == "null", (c["name"], out)
        else:
            d = golden[f"dec/{c['name']}/desc"]
            want = "%u %u %u %u %08x" % (int(d[0]), int(d[1]), int(d[2]), int(d[3]), zlib.crc32(golden[f"dec/{c['name']}/pixels"].tobytes()) & 0xFFFFFFFF)
            assert out == want, (c["name"], out, want)
