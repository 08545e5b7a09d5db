#!/usr/bin/env python
"""bench.py — Mpixels/s encode+decode of 4K RGBA frames on MI355X (BASELINE.json metric).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ...
  One process per GPU; frames shard across ranks with NO data-path collective (every image
  is coded with state reset, qoi.h:393-400,533-537); RCCL only gathers counters/timings.

A STEP = one pass of the hot path over one batch: every rank encodes its F resident
3840x2160 RGBA frames (qoimi_encode_batch) and decodes the streams back
(qoimi_decode_batch); F = 256 by default (8.5 GB of pixels; 1024 = the per-GPU shard of BASELINE
configs[4] also fits, 176 GB with workspaces).  Frames are synthetic (`photo` class of qoi_amd/synth.py),
generated on the device, distinct per frame and rank, F*33 MB >> the 256 MiB Infinity Cache, and
already resident in HBM when the timed region starts.  value = pixels round-tripped per
second over all ranks.  The round trip is verified bit-exact after the timed region.

Extra objects on the JSON line:
  roofline      dominant kernel enc_slabs: algorithmic bytes (4 B read per pixel,
                SURVEY.md §8d) / its mean launch duration measured with HIP events on the
                launch stream during the timed steps, against the 8 TB/s HBM peak.
  cpu_baseline  the unmodified reference (oracle/_ref, else our C port) timed on this
                host with qoibench.c's BENCHMARK_FN semantics on a bounded sample.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
def _latest_traffic_file() -> str:
    """The most recent committed PMC traffic summary (profiles/rNN_sM_pmc_traffic.json, written by tools/gpu_session.sh)."""
    import glob
    import re
    best, key = "", (-1, -1)
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_s*_pmc_traffic.json")):
        m = re.search(r"r(\d+)_s(\d+)_pmc_traffic\.json$", f)
        if m and (int(m.group(1)), int(m.group(2))) > key:
            best, key = f, (int(m.group(1)), int(m.group(2)))
    return best


TRAFFIC_FILE = _latest_traffic_file()


def measured_traffic(kernel_prefix: str, grid_threads: int):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes of THIS command
    (tools/gpu_session.sh: separate FETCH_SIZE / WRITE_SIZE runs; the counters cannot be collected from inside
    the process).  Returned only when the profiled launch had the same grid, i.e. the same workload; FETCH_SIZE
    is doubled as MI355X_MICROARCH.md prescribes for gfx950 (calibration in the file)."""
    try:
        doc = json.load(open(TRAFFIC_FILE))
    except OSError:
        return None, None
    for name, k in doc["kernels"].items():
        if name.startswith(kernel_prefix) and grid_threads in (k.get("grid"), -1) and "FETCH_SIZE_KB" in k and "WRITE_SIZE_KB" in k:
            nbytes = (k["FETCH_SIZE_KB"] * doc["read_correction"] + k["WRITE_SIZE_KB"]) * 1024.0
            return nbytes, f"profiles/{os.path.basename(TRAFFIC_FILE)}: {name} (2 x FETCH_SIZE + WRITE_SIZE)"
    return None, None


def cpu_baseline(kind: str, w: int, h: int, budget_s: float) -> dict:
    """Reference qoi.h on ONE host core: encode+decode of the same synthetic frames,
    timed like qoibench.c:364-376 (one warm-up discarded, malloc/free inside the region)."""
    from oracle import oracle_py
    from qoi_amd import synth
    lib = oracle_py.load_ref()
    if lib is None:
        lib = oracle_py.load_port()
    npx = w * h
    frames = [np.ascontiguousarray(synth.frame_rgba(kind, w, h, f)) for f in range(2)]
    desc = oracle_py.QoiDesc(w, h, 4, 0)
    enc_ns = dec_ns = 0
    runs = 0
    t_start = time.perf_counter()
    it = 0
    while True:
        fr = frames[it % len(frames)]
        t0 = time.perf_counter_ns()
        p, n = lib.encode_raw(fr.ctypes.data, desc)
        t1 = time.perf_counter_ns()
        q, _ = lib.decode_raw(p, n, 4)
        lib.free(q)
        t2 = time.perf_counter_ns()
        lib.free(p)
        if it > 0:                      # first run is the warm-up (qoibench.c:366-372)
            enc_ns += t1 - t0
            dec_ns += t2 - t1
            runs += 1
        it += 1
        if runs >= 2 and time.perf_counter() - t_start > budget_s:
            break
    enc_mpps = npx * runs / (enc_ns / 1000.0)
    dec_mpps = npx * runs / (dec_ns / 1000.0)
    both = npx * runs / ((enc_ns + dec_ns) / 1000.0)
    return {"value": round(both, 2), "unit": "Mpixels/s", "cores": 1, "kind": lib.kind,
            "sample": f"{runs} x {w}x{h} {kind} frames, encode+decode, 1 warm-up discarded, malloc/free timed",
            "encode_mpps": round(enc_mpps, 2), "decode_mpps": round(dec_mpps, 2)}


def cpu_baseline_all_cores(kind: str, w: int, h: int, budget_s: float) -> dict:
    """The same measurement on every host core at once ("(ii) one process per host core, all cores,
    core count printed"): one worker process per core, each encoding + decoding its own frames for `budget_s`;
    the rates add up.  Workers are fresh interpreters (no CUDA state is forked)."""
    import subprocess
    cores = os.cpu_count() or 1
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-worker", kind, str(w), str(h), str(budget_s)]
    procs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True) for _ in range(cores)]
    total = 0.0
    done = 0
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=budget_s * 6 + 120)
            total += float(json.loads(out.strip().splitlines()[-1])["value"]); done += 1
        except Exception:
            pr.kill()
    return {"value": round(total, 1), "unit": "Mpixels/s", "cores": done, "kind": "reference",
            "sample": f"{done} worker processes x {budget_s:.0f} s of {w}x{h} {kind} frames, encode+decode, malloc/free timed"}


def main() -> None:
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":            # one core's share of cpu_baseline_all_cores
        print(json.dumps(cpu_baseline(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=256, help="4K frames resident per GPU (= per step); BASELINE configs[4] is 1024 per GPU")
    ap.add_argument("--kind", default="photo", choices=["photo", "noise", "uiflat", "constant"])
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--encode-only", action="store_true", help="diagnostics: time the encoder alone (no decode, no check)")
    ap.add_argument("--no-others", action="store_true", help="skip the noise / constant / uiflat side figures")
    ap.add_argument("--no-single", action="store_true", help="skip the single-frame figure (profiling runs: keeps every launch batch-sized)")
    args = ap.parse_args()

    import torch
    from qoi_amd import api, synth
    from qoi_amd import dist as qdist

    rank, world, local = qdist.env_world()
    assert world == args.gpus or world == 1, f"launched with WORLD_SIZE={world} but --gpus {args.gpus}"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    qdist.init("nccl", dev)          # RCCL; only counters ever cross GPUs
    ctx = api.Context(local)
    stream = torch.cuda.current_stream().cuda_stream

    w, h, F = args.width, args.height, args.frames
    npx = w * h
    desc = api.QoiDesc(w, h, 4, api.QOI_SRGB)
    pstride = (npx * 4 + 255) // 256 * 256
    sstride = (api.encode_bound(w, h, 4) + 255) // 256 * 256
    pixels = torch.empty(F * pstride, dtype=torch.uint8, device=dev)
    streams = torch.empty(F * sstride, dtype=torch.uint8, device=dev)
    decoded = torch.empty(F * pstride, dtype=torch.uint8, device=dev)
    lens = torch.zeros(F, dtype=torch.int32, device=dev)
    my_frames = qdist.shard_frames(rank, world, F)      # weak scaling: F distinct frames per GPU
    ctx.synth_frames(synth.KIND_ID[args.kind], synth.DEFAULT_SEED, my_frames[0], F, w, h,
                     pixels.data_ptr(), pstride, stream)
    torch.cuda.synchronize()

    # stream lengths are data-dependent; they are constant across steps, read them once
    ctx.encode_batch(pixels.data_ptr(), pstride, desc, F, streams.data_ptr(), sstride, lens.data_ptr(), stream)
    ctx.encode_status(stream)
    sizes = [int(x) for x in lens.cpu().numpy()]
    descs = [desc] * F

    def step():
        ctx.encode_batch(pixels.data_ptr(), pstride, desc, F, streams.data_ptr(), sstride, lens.data_ptr(), stream)
        if args.encode_only:
            return
        ctx.decode_batch(streams.data_ptr(), sstride, sizes, descs, 4, decoded.data_ptr(), pstride, stream)

    def barrier():
        qdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = ctx.get_profile(stream)
    ctx.set_profiling(False)
    ctx.encode_status(stream)

    # BASELINE configs[1]: ONE 4K frame, encode + decode, device-resident (33 MB: served by the 256 MiB Infinity
    # Cache on repeat runs and bound by launch latency, not by HBM - reported next to the batch figure, never as it)
    single = None
    if not args.encode_only and not args.no_single and rank == 0:
        one = [sizes[0]]
        for _ in range(3):
            ctx.encode_batch(pixels.data_ptr(), pstride, desc, 1, streams.data_ptr(), sstride, lens.data_ptr(), stream)
            ctx.decode_batch(streams.data_ptr(), sstride, one, [desc], 4, decoded.data_ptr(), pstride, stream)
        torch.cuda.synchronize()
        reps = 20
        t1 = time.perf_counter()
        for _ in range(reps):
            ctx.encode_batch(pixels.data_ptr(), pstride, desc, 1, streams.data_ptr(), sstride, lens.data_ptr(), stream)
            ctx.decode_batch(streams.data_ptr(), sstride, one, [desc], 4, decoded.data_ptr(), pstride, stream)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / reps
        single = {"workload": f"1 x {w}x{h} RGBA frame, encode + decode, device-resident, wall clock incl. launches",
                  "ms": round(dt * 1e3, 4), "mpixels_per_s": round(npx / dt / 1e6, 1)}

    # bit-exact round trip (qoibench.c:408-417) checked outside the timed region
    ok = args.encode_only or bool(torch.equal(decoded.view(F, -1)[:, :npx * 4], pixels.view(F, -1)[:, :npx * 4]))
    dstats = ctx.decode_stats()

    # next to the headline content always report `noise` (5 B/px written: most stream traffic) and
    # `constant` (longest runs) - same batch, 3 timed steps each, rank 0 of a single-GPU run only
    other = None
    if world == 1 and not args.encode_only and not args.no_others:
        other = {}
        for kind in ("noise", "constant", "uiflat"):
            if kind == args.kind:
                continue
            ctx.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 0, F, w, h, pixels.data_ptr(), pstride, stream)
            ctx.encode_batch(pixels.data_ptr(), pstride, desc, F, streams.data_ptr(), sstride, lens.data_ptr(), stream)
            ctx.encode_status(stream)
            ksizes = [int(x) for x in lens.cpu().numpy()]
            ctx.decode_batch(streams.data_ptr(), sstride, ksizes, descs, 4, decoded.data_ptr(), pstride, stream)   # warm-up
            torch.cuda.synchronize()
            t2 = time.perf_counter()
            for _ in range(3):
                ctx.encode_batch(pixels.data_ptr(), pstride, desc, F, streams.data_ptr(), sstride, lens.data_ptr(), stream)
                ctx.decode_batch(streams.data_ptr(), sstride, ksizes, descs, 4, decoded.data_ptr(), pstride, stream)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t2) / 3
            kok = bool(torch.equal(decoded.view(F, -1)[:, :npx * 4], pixels.view(F, -1)[:, :npx * 4]))
            other[kind] = {"mpixels_per_s": round(F * npx / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 3),
                           "stream_bytes_per_px": round(sum(ksizes) / (F * npx), 4), "decode_rounds": ctx.decode_stats()["rounds"],
                           "verified_bit_exact": kok}

    # RCCL: counters only (max elapsed; summed pixels / stream bytes / verified ranks)
    elapsed, (total_px, total_stream_bytes, n_ok) = qdist.reduce_counters(
        elapsed, [float(F * npx * args.steps), float(sum(sizes)) * args.steps, float(ok)], dev)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = total_px / elapsed / 1e6
        enc_ms = sum(prof[k][0] for k in prof if k.startswith("enc_"))
        dec_ms = sum(prof[k][0] for k in prof if k.startswith("dec_"))
        # The encode kernel of the roofline: enc_slabs plus the entry-state passes that run before its second launch
        # for images the first launch could not finish on its own (flat content) - every kernel that reads pixels.
        slabs_calls = prof["enc_slabs"][1]
        slabs_ms = sum(prof[k][0] for k in ("enc_slabs", "enc_slabs_generic", "enc_slab_summary", "enc_scan_groups", "enc_scan_images") if k in prof)
        per_launch_ms = slabs_ms / max(1, slabs_calls)
        alg_bytes = F * npx * 4                           # 4 B read per pixel
        achieved = alg_bytes / (per_launch_ms * 1e-3) / 1e9 if per_launch_ms > 0 else 0.0
        # the largest kernel of the step by time, for the record: stream bytes read + 4 B written per pixel
        seg_ms = prof["dec_segments"][0] / max(1, args.steps) if "dec_segments" in prof else 0.0
        dec_bytes = F * npx * 4 + total_stream_bytes / max(1, args.steps) / world
        dec_achieved = dec_bytes / (seg_ms * 1e-3) / 1e9 if seg_ms > 0 else 0.0

        slabs = (npx + 1023) // 1024
        traffic, traffic_src = measured_traffic("qoimi::enc_slabs<4, 16, 1, 0, 1>", ((slabs + 3) // 4) * F * 256)
        # the committed PMC run is this workload iff its enc_slabs launch had this grid (same frames, same shape)
        dec_traffic, dec_traffic_src = measured_traffic("qoimi::dec_segments", -1) if traffic is not None and args.kind == "photo" else (None, None)
        out = {
            "metric": "Mpixels/s encode+decode, 4K RGBA", "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": f"batch of {F} x {w}x{h} RGBA frames per GPU per step (BASELINE configs[4]: 8192 such frames "
                                   f"over 8 GPUs = 1024 per GPU; --frames 1024 runs that shard, the default keeps a quarter of it "
                                   f"resident), encode + decode, content={args.kind}, HBM-resident, bit-exact round trip verified",
                       "frames_per_gpu": F, "width": w, "height": h, "content": args.kind,
                       "stream_bytes_per_px": round(total_stream_bytes / total_px, 4), "parallelism": f"frames sharded x{world}"},
            "verified_bit_exact": n_ok == world,
            "encode_mpps_kernels": round(F * npx * args.steps / (enc_ms * 1e3), 1) if enc_ms else None,
            "decode_mpps_kernels": round(F * npx * args.steps / (dec_ms * 1e3), 1) if dec_ms else None,
            "decode_rounds": dstats["rounds"], "decode_redo_segments": dstats["redo_segments"],
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]},
            "roofline": {"bound": "hbm", "kernel": "enc_slabs (+ entry-state passes)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": alg_bytes, "ms_per_launch": round(per_launch_ms, 4)},
            "roofline_decode": {"bound": "hbm", "kernel": "dec_segments_pair", "achieved": round(dec_achieved, 1), "peak": HBM_PEAK_GBS,
                                "unit": "GB/s", "frac": round(dec_achieved / HBM_PEAK_GBS, 4), "traffic": dec_traffic,
                                "traffic_source": dec_traffic_src, "algorithmic_bytes_per_launch": int(dec_bytes),
                                "ms_per_launch": round(seg_ms, 4),
                                "note": "bound by resident lanes x instructions per chunk (LDS colour tables), not by HBM: DESIGN.md sections 2 and 4"},
        }
        if single:
            out["single_frame"] = single
        if other:
            out["other_content"] = other
        if args.encode_only:
            out["config"]["workload"] += " [ENCODE ONLY - diagnostic run, not the benchmark]"
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.kind, w, h, args.cpu_seconds)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.kind, w, h, min(args.cpu_seconds, 6.0))
        print(json.dumps(out), flush=True)
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
