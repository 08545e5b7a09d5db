"""Test / measurement helper: point qoi_amd.api at another build of the library BEFORE its first use in this process.

The product binding (qoi_amd/api.py) loads qoi_amd/lib/libqoi_mi355x.so and nothing in the environment changes that.  Tests that
need the failure-injection hooks (QOIMI_TEST_SPIN_BOUND, QOIMI_TEST_FORCE_RECHECK_FAIL: compiled into the test flavour only,
`make -C qoi_amd/csrc TEST_HOOKS=1`) run their scenario in a child process that calls use_test_library() first; measurement tools
that compare builds call use_library(path) or honour QOIMI_TOOLS_LIB through it.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEST_LIB = os.path.join(ROOT, "qoi_amd", "lib", "libqoi_mi355x_test.so")


def use_library(path: str) -> None:
    from qoi_amd import api
    if api._lib is not None:
        raise RuntimeError("qoi_amd.api has loaded its library already; select another build before the first call")
    api.LIB_PATH = os.path.abspath(path)


def use_test_library() -> None:
    if not os.path.exists(TEST_LIB):
        raise RuntimeError(f"{TEST_LIB} not built - run `make -C qoi_amd/csrc TEST_HOOKS=1` (or __graft_entry__.build())")
    use_library(TEST_LIB)


def use_env_library() -> None:
    """Measurement tools: QOIMI_TOOLS_LIB=<path> selects the build under test (tools/measure/ab_*.sh)."""
    if os.environ.get("QOIMI_TOOLS_LIB"):
        use_library(os.environ["QOIMI_TOOLS_LIB"])
