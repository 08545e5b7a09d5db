"""bench.py's multi-rank path on a ONE-GPU box: two ranks (torch.distributed.run, started by bench.py itself) share GPU 0, counters
travel over gloo as CPU tensors (`--counter-backend gloo`).  Everything else is the code the driver runs with RCCL on N GPUs:
rank environment, frame sharding (weak: F distinct frames per rank; strong: a fixed job divided), barrier + max-over-ranks
timing, counter reduction, one JSON line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "2", "--warmup", "1", "--width", "1920", "--height", "1080", "--no-others", "--no-single", "--no-configs"]


def run_bench(extra):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + COMMON + extra, env=env, capture_output=True, text=True, timeout=900)
    return r


@pytest.mark.gpu
def test_two_ranks_weak_scaling_on_one_gpu():
    # the CPU baseline stays on (1 s budgets): an N > 1 line must carry the same objects as the N = 1 line
    r = run_bench(["--gpus", "2", "--frames", "8", "--counter-backend", "gloo", "--cpu-seconds", "1"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["counter_backend"] == "gloo"
    for key in ("roofline", "roofline_dominant", "roofline_decode_total", "cpu_baseline", "cpu_baseline_all_cores", "ms_per_step_ranks"):
        assert key in out, key
    assert out["roofline"]["frac"] > 0 and out["roofline"]["bound"] == "hbm"
    assert out["cpu_baseline"]["value"] > 0 and out["cpu_baseline"]["cores"] == 1 and out["cpu_baseline"]["kind"] in ("reference", "port")
    assert out["cpu_baseline_all_cores"]["cores"] >= 1
    assert 0 < out["ms_per_step_ranks"]["min"] <= out["ms_per_step_ranks"]["max"] == out["ms_per_step"]
    assert out["verified_bit_exact"] is True
    assert out["config"]["frames_per_gpu"] == 8
    # ranks coded disjoint frame ids 0..7 and 8..15: 16 frames, ids summing to 120
    assert out["frames_coded_all_ranks"] == 16 and out["frame_id_sum_all_ranks"] == sum(range(16))
    # value = pixels of BOTH ranks per second of the slower rank
    px = 2 * 8 * 1920 * 1080 * out["steps"]
    assert abs(out["value"] - px / (out["ms_per_step"] * 1e-3 * out["steps"]) / 1e6) / out["value"] < 0.01


@pytest.mark.gpu
def test_two_ranks_strong_scaling_on_one_gpu():
    r = run_bench(["--gpus", "2", "--frames", "4", "--scaling", "strong", "--total-frames", "12", "--counter-backend", "gloo", "--no-cpu"])
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-1500:])
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["scaling"] == "strong" and out["verified_bit_exact"] is True
    # 12 frames over 2 ranks = 6 each, in 2 passes over 4 resident frames; ids 0..11 covered exactly once
    assert out["frames_coded_all_ranks"] == 12 and out["frame_id_sum_all_ranks"] == sum(range(12))


@pytest.mark.gpu
def test_more_ranks_than_gpus_is_refused_under_rccl():
    import torch
    n = torch.cuda.device_count()
    r = run_bench(["--gpus", str(n + 1), "--frames", "2", "--no-cpu"])
    assert r.returncode != 0
    assert "GPU(s) visible" in (r.stdout + r.stderr)
