"""tools/qoibench_mi355x.py: qoibench.c's table and flags, driven here with a CPU build of the
reference in the baseline row and no GPU (the tool itself has no CPU codec)."""
import importlib.util
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ref(m):
    from oracle import oracle_py
    oracle_py.load_ref() or oracle_py.load_port()          # builds the checker libraries if needed
    here = os.path.join(ROOT, "oracle")
    path, prefix = (os.path.join(here, "_ref", "libqoiref.so"), "ref_")
    if not os.path.exists(path):
        path, prefix = os.path.join(here, "liboracle.so"), "oracle_"
    return m.RefCodec(path, prefix)


def _load():
    spec = importlib.util.spec_from_file_location("qoibench_mi355x", os.path.join(ROOT, "tools", "qoibench_mi355x.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_table_format_and_totals(tmp_path):
    m = _load()
    lines = []
    rc = m.main(["1", str(tmp_path), "--synth", "1", "--nowarmup"], out=lambda s="": lines.extend(str(s).split("\n")),
                ref=_ref(m), use_gpu=False)
    assert rc == 0
    text = "\n".join(lines)
    head = "          decode ms   encode ms   decode mpps   encode mpps   size kb    rate"      # qoibench.c:339
    n_img = len(__import__("qoi_amd.synth", fromlist=["KINDS"]).KINDS)      # --synth 1: one image per content class
    assert text.count(head) == n_img + 2                  # the images, directory total, grand total
    rows = [l for l in lines if l.startswith("qoi-ref:")]
    assert len(rows) == n_img + 2
    for r in rows:                                        # "%s   %8.1f    %8.1f      %8.2f      %8.2f  %8ld   %4.1f%%"
        assert re.match(r"^qoi-ref:\s+ +\d+\.\d +\d+\.\d +\d+\.\d\d +\d+\.\d\d +\d+ +\d+\.\d%$", r), r
    assert "# Grand total for" in text and "## Total for" in text


def test_onlytotals_and_flags(tmp_path):
    m = _load()
    lines = []
    m.main(["1", str(tmp_path), "--synth", "1", "--onlytotals", "--noencode", "--nowarmup"],
           out=lambda s="": lines.extend(str(s).split("\n")), ref=_ref(m), use_gpu=False)
    rows = [l for l in lines if l.startswith("qoi-ref:")]
    assert len(rows) == 2                                 # directory total + grand total only
    assert all(float(r.split()[2]) == 0.0 for r in rows)  # --noencode: encode ms column stays 0


@pytest.mark.gpu
def test_gpu_rows_on_a_512_png(tmp_path):
    """BASELINE configs[0] on the GPU rows: one 512 x 512 RGBA PNG, qoibench.c's verification (:408-417) and table
    (:335-360) with the MI355X library in the `qoi` rows - drop-in (host pointers) and device-resident - and the reference
    CPU build beside them.  Sizes must agree between the rows: the GPU stream is the reference's stream."""
    import torch  # noqa: F401
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import png_io
    from qoi_amd import synth
    m = _load()
    px = synth.frame_rgba("photo", 512, 512, 3).reshape(512, 512, 4)
    (tmp_path / "plumbing_512.png").write_bytes(png_io.write_png(px))
    px3 = np.ascontiguousarray(synth.frame_rgba("uiflat", 200, 120, 1).reshape(120, 200, 4)[:, :, :3])
    (tmp_path / "rgb_200x120.png").write_bytes(png_io.write_png(px3))
    lines = []
    rc = m.main(["3", str(tmp_path)], out=lambda s="": lines.extend(str(s).split("\n")), ref=_ref(m), use_gpu=True)
    assert rc == 0
    rows = {k: [l for l in lines if l.startswith(k)] for k in ("qoi-mi355x:", "qoi-dev:", "qoi-ref:")}
    assert all(len(v) == 2 + 2 for v in rows.values()), {k: len(v) for k, v in rows.items()}     # two images, directory total, grand total
    # the directory in ONE qoimi_encode_images / qoimi_decode_batch call (a 4-channel and a 3-channel image of different shapes): in the two totals
    batch = [l for l in lines if l.startswith("qoi-batch:")]
    assert len(batch) == 2 and batch[0].split()[-2:] == rows["qoi-ref:"][2].split()[-2:], (batch, rows["qoi-ref:"])
    assert float(batch[0].split()[3]) > 0 and float(batch[0].split()[4]) > 0
    for a, b, c in zip(rows["qoi-mi355x:"], rows["qoi-dev:"], rows["qoi-ref:"]):
        assert a.split()[-2:] == c.split()[-2:] == b.split()[-2:], (a, b, c)      # size kb and rate: byte-identical streams
        for r in (a, b, c):
            assert float(r.split()[3]) > 0 and float(r.split()[4]) > 0           # decode mpps, encode mpps measured (the ms columns print 0.0 below 50 us)
