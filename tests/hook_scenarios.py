"""Scenarios that need the library's failure-injection hooks, run as a child process of the GPU tests on the TEST flavour of the
library (libqoi_mi355x_test.so: make -C qoi_amd/csrc TEST_HOOKS=1).  usage: python tests/hook_scenarios.py <scenario> [args]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
os.environ["QOIMI_TUNING"] = "1"

import libsel  # noqa: E402

libsel.use_test_library()

import torch  # noqa: E402,F401  (first: the library binds to the HIP runtime torch loaded)
from gpu_util import DeviceBatch  # noqa: E402
from oracle import oracle_py  # noqa: E402
from qoi_amd import api, synth  # noqa: E402

oracle = oracle_py.load_ref() or oracle_py.load_port()


def spin_bound(n, w, h):
    """QOIMI_TEST_SPIN_BOUND=1 with the tree placed by workgroup index (the form whose waits are a bet) and with look-back placement:
    the wait gives up, qoimi_encode_status encodes again order-free."""
    os.environ["QOIMI_TEST_SPIN_BOUND"] = "1"
    os.environ["QOIMI_ENC_TREE_TICKET"] = "0"
    c = api.Context(0)
    del os.environ["QOIMI_TEST_SPIN_BOUND"], os.environ["QOIMI_ENC_TREE_TICKET"]
    b = DeviceBatch(c, w, h, 4, n)
    for i in range(n):
        c.synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 300 + i, 1, w, h, b.pixels.data_ptr() + i * b.pixel_stride, b.pixel_stride, b.stream)
    lens = b.encode()                 # encode_batch + encode_status (raises on an error status)
    for i in range(n):
        assert b.stream_bytes(i, lens[i]) == oracle.encode(synth.frame_rgba("photo", w, h, 300 + i), w, h, 4), i
    r = c.encode_retries()
    assert r >= 1, "the bounded wait never gave up: the hook did not act"
    print(f"retries {r}")
    c.close()


def recheck_fail():
    os.environ.update({"QOIMI_ENC_RECHECK_EVERY": "1", "QOIMI_TEST_FORCE_RECHECK_FAIL": "1"})
    c = api.Context(0)
    w, h = 640, 360
    b = DeviceBatch(c, w, h, 4, 2)
    frames = [synth.frame_rgba(k, w, h, 11 + i) for i, k in enumerate(("photo", "uiflat"))]
    for i, f in enumerate(frames):
        b.upload(i, f)
    want = [oracle.encode(f, w, h, 4) for f in frames]
    lens = b.encode()                                   # call 1: launches the first repeat
    assert c.encode_suspect_calls() == 0
    assert [b.stream_bytes(i, lens[i]) for i in range(2)] == want
    torch.cuda.synchronize()
    # call 2 notices the (forced) failure: it still succeeds, byte-identical, now with the order-independent probe
    c.encode_batch(b.pixels.data_ptr(), b.pixel_stride, b.desc, b.n, b.streams.data_ptr(), b.stream_stride, b.lens.data_ptr(), b.stream)
    assert c.encode_suspect_calls() == 1                # call 1 was made with the suspect probe
    try:
        c.encode_status(b.stream)
        raise AssertionError("the failed repeat was not reported")
    except api.QoiError as e:
        assert "self-test failed" in str(e)
    c.encode_status(b.stream)                           # reported once
    lens = b.lens.cpu().numpy()
    assert [b.stream_bytes(i, lens[i]) for i in range(2)] == want
    lens = b.encode()                                   # call 3: no further repeats, nothing more to report
    assert c.encode_suspect_calls() == 1
    assert [b.stream_bytes(i, lens[i]) for i in range(2)] == want
    c.close()
    print("ok")


if __name__ == "__main__":
    name, args = sys.argv[1], [int(a) for a in sys.argv[2:]]
    {"spin_bound": spin_bound, "recheck_fail": recheck_fail}[name](*args)
