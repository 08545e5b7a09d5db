"""The reference's robustness tool is libFuzzer + AddressSanitizer on qoi_decode (qoifuzz.c:9-12,20-32).  Here the same
harness shape runs against the drop-in qoi_decode with the HOST shim built with ASan + UBSan + libFuzzer coverage
(tests/fuzz/Makefile), and every input is also decoded by the unmodified reference in the same process and compared
(tests/fuzz/qoi_fuzz_diff.c).  This test runs a short campaign; a 20 000-input run is kept under profiles/."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_bin", "qoi_fuzz_diff")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:allocator_may_return_null=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")


def test_fuzz_harness_sources_present():
    for f in ("qoi_fuzz_diff.c", "Makefile", "make_corpus.py"):
        assert os.path.exists(os.path.join(ROOT, "tests", "fuzz", f))


@pytest.mark.gpu
def test_differential_fuzz_under_asan(tmp_path):
    if not os.path.exists(BIN):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tests", "fuzz")], check=True, capture_output=True, timeout=900)
    corpus = tmp_path / "corpus"
    subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fuzz", "make_corpus.py"), str(corpus)], check=True, capture_output=True)
    r = subprocess.run([BIN, "-runs=2500", "-rss_limit_mb=8192", "-max_len=8192", "-seed=20260923", "-timeout=60", "-print_final_stats=1", str(corpus)],
                       env=ENV, capture_output=True, text=True, timeout=900)
    tail = r.stderr[-3000:]
    assert r.returncode == 0, tail
    assert "MISMATCH" not in r.stderr and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, tail
    assert "decoded by both" in r.stderr, tail
