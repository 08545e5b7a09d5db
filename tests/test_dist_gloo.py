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


WORKER8 = textwrap.dedent("""
    import json, os, sys, time
    sys.path.insert(0, %r)
    import torch
    from qoi_amd import dist as qdist
    rank, world, local = qdist.env_world()
    qdist.init("gloo")
    # BASELINE configs[4], literally: 8192 x 4K frames over 8 GPUs.  Weak: 1024 frames per rank whatever the world size; strong: the
    # 8192 frames divided.  No frame is coded here (the arithmetic of the sharding and of the counters is what runs on eight ranks).
    TOTAL, F, npx = 8192, 1024, 3840 * 2160
    weak = qdist.shard_frames(rank, world, F)
    strong = qdist.shard_range(TOTAL, rank, world)
    odd = qdist.shard_range(8191, rank, world)                # a total the ranks do not divide
    elapsed = 0.050 + 0.001 * ((rank * 5) %% world)            # every rank its own clock: max and min must come from different ranks
    peak = float((150 + rank) << 30)                            # per-rank peak device bytes, as bench.py gathers them
    mx, (px, frames, fsum, fsum_strong, n_odd, peak_sum) = qdist.reduce_counters(
        elapsed, [float(len(weak) * npx), float(len(weak)), float(sum(weak)), float(sum(strong)), float(len(odd)), peak])
    mn = qdist.reduce_min(elapsed)
    peak_max, _ = qdist.reduce_counters(peak, [])
    if rank == 0:
        print(json.dumps({"world": world, "max": mx, "min": mn, "px": px, "frames": frames, "fsum": fsum, "fsum_strong": fsum_strong,
                          "n_odd": n_odd, "peak_sum": peak_sum, "peak_max": peak_max,
                          "strong_first": [strong.start, strong.stop], "weak_first": [weak[0], weak[-1]]}))
    qdist.barrier()
    torch.distributed.destroy_process_group()
""") % ROOT


def test_eight_rank_gloo_configs4_arithmetic(tmp_path):
    """World size 8 on the CPU with the literal numbers of BASELINE configs[4] (the driver's 8-GPU run cannot be rehearsed on hardware
    here): weak scaling = 1024 frames per rank, 8192 in total, disjoint frame ids; strong scaling = shard_range over 8192 (and over a
    total the ranks do not divide); the reduced counters - pixels, frames, frame-id sums, max / min elapsed, per-rank peak device bytes
    (max and sum) - are what bench.py prints at N = 8."""
    script = tmp_path / "worker8.py"
    script.write_text(WORKER8)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8",
           "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["world"] == 8
    assert out["frames"] == 8192 and out["px"] == 8192 * 3840 * 2160
    assert out["fsum"] == sum(range(8192)) == out["fsum_strong"]            # both partitions cover frames 0..8191 exactly once
    assert out["n_odd"] == 8191
    assert out["strong_first"] == [0, 1024] and out["weak_first"] == [0, 1023]
    assert abs(out["max"] - 0.057) < 1e-9 and abs(out["min"] - 0.050) < 1e-9
    assert out["peak_max"] == float(157 << 30) and out["peak_sum"] == float(sum((150 + r) << 30 for r in range(8)))
    # eight ranks of the headline shard at once: 8 x 170 GB of 288 GB each is per GPU, not per node - every rank's peak must fit ITS device
    assert out["peak_max"] < 288e9
