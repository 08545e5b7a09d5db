"""Re-entrancy of the drop-in entry points (qoi.h:339,357-362,489-495: no state between calls, callable from any thread):
several threads encode and decode different images at the same time through the C-ABI (ctypes releases the GIL for the
duration of a call); every result must equal the oracle's."""
import threading

import numpy as np
import pytest

from qoi_amd import synth

pytestmark = pytest.mark.gpu


def test_concurrent_encode_decode(ref, port):
    import torch  # noqa: F401
    from qoi_amd import api
    oracle = ref or port
    kinds = ("photo", "noise", "uiflat", "constant")
    jobs = []
    for t in range(8):
        w, h = 320 + 37 * t, 200 + 11 * t
        px = synth.frame_rgba(kinds[t % 4], w, h, t)
        jobs.append((w, h, px, oracle.encode(px, w, h, 4)))
    errors = []
    start = threading.Barrier(len(jobs))

    def work(i):
        w, h, px, want = jobs[i]
        try:
            start.wait()
            for it in range(6):
                s = api.qoi_encode(px, api.QoiDesc(w, h, 4, 0))
                if s != want:
                    errors.append((i, it, "encode"))
                got, d = api.qoi_decode(want, 4 if it % 2 else 0)
                if got is None or not np.array_equal(got, px.reshape(-1)) or (d.width, d.height) != (w, h):
                    errors.append((i, it, "decode"))
        except Exception as e:                      # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=work, args=(i,)) for i in range(len(jobs))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_concurrent_device_frames_under_waiting_placements():
    """tests/stress_threads.py, short: six threads with their own contexts and streams encode 4K frames at the same time under the tree,
    look-back and order-free placements - kernels whose sets wait on lower-numbered ones compete for the workgroup slots; every stream
    equals the reference's, no spin bound trips (qoimi_encode_status).  Longer runs: profiles/r04_stress_threads.txt."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_threads.py"), "--threads", "6", "--calls", "12"],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "no placement wait gave up" in r.stdout


@pytest.mark.gpu
def test_single_frames_beside_whole_batches():
    """Round 5's review, 3c: three contexts on three streams run single-frame encodes (the library's default placement: tickets) WHILE a
    fourth context round-trips whole batches on a stream of its own - thousands of workgroups of the batch's kernels between the small
    calls' workgroups.  Every stream equals the reference's, every batch hashes like the first, no placement wait gives up
    (qoimi_encode_retries stays 0).  The 1024-frame campaign: profiles/r06_stress_batches.txt."""
    import os
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "stress_threads.py"), "--threads", "3", "--calls", "60",
                        "--batch-frames", "96", "--default-placement"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "no placement wait gave up" in r.stdout and "batches of 96 frames" in r.stdout
