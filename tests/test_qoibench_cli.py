"""tools/qoibench_mi355x.py: qoibench.c's table and flags, driven here with a CPU build of the
reference in the baseline row and no GPU (the tool itself has no CPU codec)."""
import importlib.util
import os
import re

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
    assert text.count(head) == 4 + 2                      # four images, directory total, grand total
    rows = [l for l in lines if l.startswith("qoi-ref:")]
    assert len(rows) == 6
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
