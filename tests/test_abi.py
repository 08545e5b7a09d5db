"""CPU-only: the C-ABI library loads and exports every symbol include/qoi_mi355x.h declares;
argument validation that needs no GPU behaves like the reference (qoi.h:364-372, 497-503)."""
import ctypes
import os
import re

import numpy as np
import pytest

from qoi_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(api.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    return api.load_library()


def test_header_symbols_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "qoi_mi355x.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(qoi_[a-z]+|qoimi_[a-z_]+)\s*\(", hdr))
    assert declared == set(api.EXPORTS), declared ^ set(api.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_nothing_else_is_exported():
    """-fvisibility=hidden (qoi_amd/csrc/Makefile): the dynamic symbol table of every flavour holds the functions the header declares
    and nothing of the implementation (no mangled qoimi:: kernels or launchers, no helper classes)."""
    import subprocess
    for flavour in ("libqoi_mi355x.so", "libqoi_mi355x_nostdio.so", "libqoi_mi355x_test.so"):
        path = os.path.join(ROOT, "qoi_amd", "lib", flavour)
        if not os.path.exists(path):
            continue
        syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
        names = {l.split()[-1] for l in syms.splitlines() if l.strip()}
        extra = {n for n in names if not re.fullmatch(r"qoi_[a-z]+|qoimi_[a-z_]+", n)}
        assert not extra, (flavour, sorted(extra)[:10])
        want = set(api.EXPORTS) - ({"qoi_write", "qoi_read"} if "nostdio" in flavour else set())
        assert names == want, (flavour, names ^ want)


def test_environment_knobs_are_gated(lib):
    """The product library reads its tuning knobs only under QOIMI_TUNING=1 and holds no failure-injection hook at all: the strings
    of the test hooks occur in the test flavour only."""
    prod = open(os.path.join(ROOT, "qoi_amd", "lib", "libqoi_mi355x.so"), "rb").read()
    assert b"QOIMI_TUNING" in prod and b"QOIMI_TEST_SPIN_BOUND" not in prod and b"QOIMI_TEST_FORCE_RECHECK_FAIL" not in prod
    test = os.path.join(ROOT, "qoi_amd", "lib", "libqoi_mi355x_test.so")
    if os.path.exists(test):
        assert b"QOIMI_TEST_SPIN_BOUND" in open(test, "rb").read()
    assert "QOIMI_LIB" not in open(os.path.join(ROOT, "qoi_amd", "api.py")).read()
    # ... and no kernel file asks the environment anything; the host shim asks for three documented settings, QOIMI_TUNING, and - inside
    # the gate or the test-hook block - the knobs
    for f in ("qoi_encode.hip", "qoi_decode.hip", "qoi_synth.hip"):
        assert "getenv" not in open(os.path.join(ROOT, "qoi_amd", "csrc", f)).read(), f
    host = open(os.path.join(ROOT, "qoi_amd", "csrc", "qoi_host.hip")).read()
    gate = host.index('getenv("QOIMI_TUNING")')
    outside = [m for m in re.findall(r'getenv\("(QOIMI_[A-Z0-9_]+)"\)', host[:gate])]
    assert sorted(outside) == ["QOIMI_ENCODE_TIGHT_BUFFER", "QOIMI_ENC_PROBE"], outside
    assert re.findall(r'getenv\("(QOIMI_[A-Z0-9_]+)"\)', host[host.index("static qoimi_ctx* thread_ctx()"):]) == ["QOIMI_DEVICE"]


def test_desc_layout():
    assert ctypes.sizeof(api.QoiDesc) == 12
    assert (api.QoiDesc.width.offset, api.QoiDesc.height.offset,
            api.QoiDesc.channels.offset, api.QoiDesc.colorspace.offset) == (0, 4, 8, 9)


def test_encode_bound(lib):
    assert api.encode_bound(3840, 2160, 4) == 3840 * 2160 * 5 + 22      # qoi.h:374-376
    assert api.encode_bound(0, 4, 4) == 0
    assert api.encode_bound(20000, 20000, 4) == 0                       # qoi.h:369
    assert api.encode_bound(16384, 16384, 4) == 16384 * 16384 * 5 + 22


def test_argument_rejections_need_no_gpu(lib):
    px = np.zeros(64, dtype=np.uint8)
    for w, h, ch, cs in [(0, 4, 4, 0), (4, 0, 4, 0), (4, 4, 2, 0), (4, 4, 5, 0), (4, 4, 4, 2), (20000, 20000, 4, 0)]:
        assert api.qoi_encode(px, api.QoiDesc(w, h, ch, cs)) is None
    out, _ = api.qoi_decode(b"qoif" + b"\0" * 10, 4)                     # size < 22
    assert out is None
    out, _ = api.qoi_decode(b"qoif" + b"\0" * 30, 5)                     # bad channels argument
    assert out is None
    out, d = api.qoi_decode(b"qoig" + bytes([0, 0, 0, 2, 0, 0, 0, 3, 4, 0]) + b"\0" * 9, 4)   # bad magic
    assert out is None and (d.width, d.height, d.channels) == (2, 3, 4)  # desc filled before validation


def test_no_cpu_fallback_without_gpu(lib):
    """On a box without a GPU the codec must FAIL, not silently compute on the CPU."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    px = np.zeros(4 * 4 * 4, dtype=np.uint8)
    assert api.qoi_encode(px, api.QoiDesc(4, 4, 4, 0)) is None
    with pytest.raises(api.QoiError):
        api.Context(0)


def test_python_wrapper_rejects_short_buffers():
    """qoi_amd.api.qoi_encode / qoi_write check what the C functions cannot: that the pixel buffer is as long as the
    descriptor says.  No GPU is touched: the check comes first."""
    import numpy as np
    from qoi_amd import api
    short = np.zeros(100, dtype=np.uint8)
    assert api.qoi_encode(short, api.QoiDesc(64, 64, 4, 0)) is None
    assert api.qoi_write("/tmp/qoi_mi355x_never_written.qoi", short, api.QoiDesc(64, 64, 3, 0)) == 0


def test_no_stdio_flavour_lacks_the_file_functions():
    """qoi.h:51-58,592: a build with QOI_NO_STDIO has no qoi_write / qoi_read.  `make -C qoi_amd/csrc NO_STDIO=1` (run by
    __graft_entry__.build()) gives libqoi_mi355x_nostdio.so: qoi_encode / qoi_decode and the device API, nothing of stdio."""
    import subprocess
    path = os.path.join(ROOT, "qoi_amd", "lib", "libqoi_mi355x_nostdio.so")
    if not os.path.exists(path):
        pytest.skip("libqoi_mi355x_nostdio.so not built")
    syms = subprocess.run(["nm", "-D", "--defined-only", path], capture_output=True, text=True, check=True).stdout
    names = {l.split()[-1] for l in syms.splitlines() if l.strip()}
    assert {"qoi_encode", "qoi_decode", "qoimi_encode_batch", "qoimi_decode_batch"} <= names
    assert not ({"qoi_write", "qoi_read"} & names)
