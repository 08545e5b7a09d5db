"""N>1 path on CPU: two gloo ranks shard a frame batch with no data-path collective and
gather counters exactly like bench.py does over RCCL.  The per-rank 'codec' here is the
oracle (test infrastructure); what is under test is the sharding + counter reduction of
qoi_amd/dist.py."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import numpy as np
    import torch
    from qoi_amd import dist as qdist, synth
    from oracle import oracle_py
    rank, world, local = qdist.env_world()
    qdist.init("gloo")
    port = oracle_py.load_port()
    F, w, h = 3, 96, 64
    frames = qdist.shard_frames(rank, world, F)
    npx = w * h
    t0 = time.perf_counter()
    nbytes = 0; ok = 1.0
    for f in frames:
        px = synth.frame_rgba("photo", w, h, f)
        s = port.encode(px, w, h, 4)
        back, _ = port.decode(s, 4)
        ok *= float(np.array_equal(back, px.reshape(-1)))
        nbytes += len(s)
    qdist.barrier()
    elapsed = time.perf_counter() - t0 + 0.001 * rank
    mx, (tot_px, tot_bytes, n_ok, fsum) = qdist.reduce_counters(elapsed, [F * npx, nbytes, ok, float(sum(frames))])
    # strong-scaling helper: block partition covers every item exactly once
    cover = sorted(i for r in range(world) for i in qdist.shard_range(11, r, world))
    if rank == 0:
        print(json.dumps({"world": world, "max_elapsed_ge_own": mx >= elapsed - 1e-9, "tot_px": tot_px,
                          "tot_bytes": tot_bytes, "n_ok": n_ok, "fsum": fsum, "cover": cover}))
    qdist.barrier()
    torch.distributed.destroy_process_group()
""") % ROOT


def test_two_rank_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    # single-process truth
    sys.path.insert(0, ROOT)
    from oracle import oracle_py
    from qoi_amd import synth
    port = oracle_py.load_port()
    want_bytes = sum(len(port.encode(synth.frame_rgba("photo", 96, 64, f), 96, 64, 4)) for f in range(6))
    assert out["world"] == 2 and out["max_elapsed_ge_own"]
    assert out["tot_px"] == 6 * 96 * 64 and out["tot_bytes"] == want_bytes
    assert out["n_ok"] == 2.0 and out["fsum"] == sum(range(6))      # ranks own disjoint frames 0..5
    assert out["cover"] == list(range(11))
