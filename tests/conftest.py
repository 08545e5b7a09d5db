import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

# The library looks at its measurement / test knobs (QOIMI_ENC_*, QOIMI_DEC_*, QOIMI_SEG_BYTES ...) only under QOIMI_TUNING=1
# (qoi_host.hip: qoimi_ctx_create); the tests that select kernel paths with them run with it.
os.environ["QOIMI_TUNING"] = "1"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    path = os.path.join(ROOT, "tests", "golden", "qoi_golden.npz")
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def port():
    from oracle import oracle_py
    return oracle_py.load_port()


@pytest.fixture(scope="session")
def ref():
    """The unmodified reference build, or None where oracle/_ref is absent."""
    from oracle import oracle_py
    return oracle_py.load_ref()


@pytest.fixture(scope="session")
def encoded_streams(golden):
    """encode-case name -> reference stream (from the golden file)."""
    return {k.split("/")[1]: golden[k].tobytes() for k in golden if k.startswith("enc/") and k.endswith("/stream")}
