#!/usr/bin/env python
"""bench.py — Mpixels/s encode+decode of 4K RGBA frames on MI355X (BASELINE.json metric).

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched by the driver as  python -m torch.distributed.run --nproc-per-node N ...
  One process per GPU; frames shard across ranks with NO data-path collective (every image
  is coded with state reset, qoi.h:393-400,533-537); RCCL only gathers counters/timings.

A STEP = one pass of the hot path over one batch: every rank encodes its F resident
3840x2160 RGBA frames (qoimi_encode_batch) and decodes the streams back
(qoimi_decode_batch); F = 1024 by default = one GPU's shard of BASELINE configs[4] (8192 frames over
8 GPUs; 34 GB of pixels, about 190 GB with streams, output and workspaces).  Frames are synthetic
(`photo` class of qoi_amd/synth.py), generated on the device, distinct per frame and rank, and
already resident in HBM when the timed region starts.  value = pixels round-tripped per
second over all ranks.  After the timed region the round trip is verified bit-exact on the whole
batch and four of its streams are checked against the REFERENCE codec (byte-identical to its
encoder's, decoded by its decoder to the source pixels).

--scaling weak (default): F frames per GPU whatever N is.  --scaling strong: the 8192 frames of
configs[4] in total, 8192/N per GPU, processed as passes over at most F resident frames.
`--gpus N` without a torch.distributed.run environment starts the N ranks itself.

Extra objects on the JSON line (rank 0, N = 1 unless noted):
  roofline             enc_sets, the kernel the north star's 50 % target is stated on: algorithmic bytes
                       (4 B read per pixel) / its mean launch duration measured with HIP
                       events on the launch stream during the timed steps, against the 8 TB/s HBM peak
  roofline_dominant    the same for the kernel that takes the largest share of the step (dec_segments_rec)
  roofline_encode_total / roofline_decode_total   whole qoimi_encode_batch / qoimi_decode_batch calls
  single_frame         BASELINE configs[1], with its own encode roofline
  encode_1080p_batch   BASELINE configs[2]: 1024 x 1920x1080, encode only
  single_16k           BASELINE configs[3]: one 16384 x 16384 image, encode + decode
  other_content        the batch with noise / constant / uiflat / photo_hard (2.1 B/px, LUMA / RGB heavy) / sprite_alpha (soft alpha edges)
                       content: encode and decode timed apart, both against the roofline, 64 frames per class hashed against the reference encoder
  rgb_input            a quarter of the batch as 3-channel input and output
  cpu_baseline         the unmodified reference (oracle/_ref, else our C port) timed on this
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
    """The most recent committed PMC traffic summary (profiles/rNN_sM_pmc_traffic.json, written by tools/measure/session.sh)."""
    import glob
    import re
    best, key = "", (-1, -1)
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_s*_pmc_traffic.json")):
        m = re.search(r"r(\d+)_s(\d+)_pmc_traffic\.json$", f)
        if m and (int(m.group(1)), int(m.group(2))) > key:
            best, key = f, (int(m.group(1)), int(m.group(2)))
    return best


TRAFFIC_FILE = _latest_traffic_file()


def _latest_kernel_stats_file() -> str:
    """The most recent committed rocprofv3 --kernel-trace --stats summary (profiles/rNN_sM_kernel_stats.csv)."""
    import glob
    import re
    best, key = "", (-1, -1)
    for f in glob.glob(os.path.join(ROOT, "profiles", "r*_s*_kernel_stats.csv")):
        m = re.search(r"r(\d+)_s(\d+)_kernel_stats\.csv$", f)
        if m and (int(m.group(1)), int(m.group(2))) > key:
            best, key = f, (int(m.group(1)), int(m.group(2)))
    return best


def rocprof_average_ms(kernel_substr: str):
    """Average launch duration (ms) of the kernel whose name contains `kernel_substr` in the latest committed rocprofv3 summary, and
    the file's name - an EARLIER session's profile of this command (1024 frames per launch), shown beside the HIP-event figure of the
    run at hand so that the band between the two clocks is visible in one line."""
    import csv
    f = _latest_kernel_stats_file()
    if not f:
        return None, None
    try:
        best = None
        for r in csv.DictReader(open(f)):
            if kernel_substr in r["Name"] and (best is None or float(r["TotalDurationNs"]) > float(best["TotalDurationNs"])):
                best = r
        if best is None:
            return None, None
        return float(best["AverageNs"]) / 1e6, "profiles/" + os.path.basename(f)
    except (OSError, KeyError, ValueError):
        return None, None


def measured_traffic(kernel_prefix: str, scale: float = 1.0):
    """HBM bytes per launch of the dominant kernel, EXTRAPOLATED from the latest committed rocprofv3 PMC passes of this
    command (tools/measure/session.sh: separate FETCH_SIZE / WRITE_SIZE runs of `bench.py --no-cpu ...`; the counters cannot
    be collected from inside the process): the per-launch figure of the kernel instantiation named in the file, scaled
    by the frames per launch.  It is a figure of that session's build (the file names its commit where the session
    recorded one), not of the run at hand - the source string says so.  FETCH_SIZE is doubled as
    MI355X_MICROARCH.md prescribes for gfx950 (calibration in the file)."""
    try:
        doc = json.load(open(TRAFFIC_FILE))
    except OSError:
        return None, None
    for name, k in doc["kernels"].items():
        # the exact instantiation, template arguments included (files of earlier sessions name it without the round-5 arguments)
        if name in (kernel_prefix, kernel_prefix.replace(", false, false>", ", false>"), kernel_prefix.replace(", false, false>", ">")) and "FETCH_SIZE_KB" in k and "WRITE_SIZE_KB" in k:
            nbytes = (k["FETCH_SIZE_KB"] * doc["read_correction"] + k["WRITE_SIZE_KB"]) * 1024.0 * scale
            note = "" if scale == 1.0 else f", x {scale:g}: the PMC passes ran {doc.get('frames', '?')} frames per launch"
            return nbytes, (f"profiles/{os.path.basename(TRAFFIC_FILE)} (an earlier session's PMC passes, build {doc.get('commit', 'not recorded')}; "
                            f"not collected in this run): {name} (2 x FETCH_SIZE + WRITE_SIZE{note})")
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


def spawn_ranks(n: int) -> int:
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks (one per GPU, RCCL over 127.0.0.1)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd)


def check_against_reference(torch, pixels, pstride, streams, sstride, sizes, w, h, frames) -> dict:
    """The north star's criterion on streams of the benchmark batch itself, outside the timed region: the REFERENCE
    decoder turns our stream back into the source pixels, and (stronger) the stream equals the reference encoder's."""
    from oracle import oracle_py
    lib = oracle_py.load_ref()
    kind = "reference"
    if lib is None:
        lib, kind = oracle_py.load_port(), "port"
    npx = w * h
    ident = rt = True
    for f in frames:
        px = pixels[f * pstride:f * pstride + npx * 4].cpu().numpy()
        mine = streams[f * sstride:f * sstride + sizes[f]].cpu().numpy().tobytes()
        ident = ident and mine == lib.encode(px, w, h, 4)
        back, _ = lib.decode(mine, 4)
        rt = rt and back is not None and np.array_equal(back, px)
    return {"checker": kind, "frames": list(frames), "streams_byte_identical": bool(ident), "reference_decoder_round_trips": bool(rt)}


def hash_check_against_reference(torch, ctx, pixels, pstride, streams, sstride, lens, F, w, h, channels, stream, n_sample=64) -> dict:
    """Byte identity of a whole batch without moving it: the device hashes EVERY stream (qoimi_hash_streams), the reference
    encoder codes n_sample frames spread over the batch on the host cores (outside every timed region) and its streams are
    hashed with the same function (synth.stream_hash64): equal 64-bit hashes on all of them."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle_py
    from qoi_amd import synth
    lib = oracle_py.load_ref()
    kind = "reference"
    if lib is None:
        lib, kind = oracle_py.load_port(), "port"
    hashes = torch.zeros(F, dtype=torch.int64, device=pixels.device)
    ctx.hash_streams(streams.data_ptr(), sstride, lens.data_ptr(), F, hashes.data_ptr(), stream)
    torch.cuda.synchronize()
    mine = hashes.cpu().numpy().view(np.uint64)
    frames = sorted({int(round(x)) for x in np.linspace(0, F - 1, min(n_sample, F))})
    npx = w * h

    def ref_hash(f):
        px = pixels[f * pstride:f * pstride + npx * channels].cpu().numpy()
        return synth.stream_hash64(lib.encode(px, w, h, channels))
    with ThreadPoolExecutor(max_workers=min(32, os.cpu_count() or 1)) as ex:     # ctypes releases the GIL inside the reference's qoi_encode
        want = list(ex.map(ref_hash, frames))
    bad = [f for f, hw in zip(frames, want) if int(mine[f]) != hw]
    return {"checker": kind, "frames_hashed_on_device": F, "frames_checked_against_reference": len(frames), "mismatches": bad[:8],
            "all_equal": not bad, "device_hashes": mine}


def equal_batches(torch, a, b, F, stride, nbytes) -> bool:
    """decoded == source over the first nbytes of every frame, 64 frames at a time (no batch-sized temporaries)."""
    av, bv = a.view(F, stride), b.view(F, stride)
    for lo in range(0, F, 64):
        if not bool(torch.equal(av[lo:lo + 64, :nbytes], bv[lo:lo + 64, :nbytes])):
            return False
    return True


def dropin_path(torch, api, ctx, pixels, w, h, dev) -> dict:
    """The default (qoi_encode returns the reference's worst-case allocation, qoi.h:374-379) and, from a fresh thread - a calling thread's
    context reads the setting when it is created - the opt-in tight buffer (QOIMI_ENCODE_TIGHT_BUFFER=1) beside it."""
    import threading
    out = _dropin_path(torch, api, ctx, pixels, w, h, dev)
    res = {}

    def run():
        os.environ["QOIMI_ENCODE_TIGHT_BUFFER"] = "1"
        try:
            torch.cuda.set_device(dev)
            res["tight"] = _dropin_path(torch, api, ctx, pixels, w, h, dev)
        except Exception as e:                                          # a report, not a gate
            res["tight"] = {"error": repr(e)}
        finally:
            os.environ.pop("QOIMI_ENCODE_TIGHT_BUFFER", None)
    t = threading.Thread(target=run)
    t.start(); t.join()
    tight = res.get("tight", {})
    out["encode_buffer"] = "the reference's capacity, w*h*(channels+1)+22 bytes (qoi.h:374-379), only the pages the stream touches populated"
    out["tight_buffer_opt_in"] = {k: tight.get(k) for k in ("encode_ms", "results_kept_ms", "encode_mpixels_per_s", "round_trip_exact", "error") if k in tight}
    out["tight_buffer_opt_in"]["note"] = "QOIMI_ENCODE_TIGHT_BUFFER=1: qoi_encode's malloc sized by the calling thread's previous stream (+ 1/8)"
    return out


def _dropin_path(torch, api, ctx, pixels, w, h, dev) -> dict:
    """qoi_encode / qoi_decode of the C ABI (include/qoi_mi355x.h, = qoi.h:278/289) on malloc'ed host memory, one 4K frame."""
    import ctypes
    lib = api.load_library()
    npx = w * h
    host_px = pixels[:npx * 4].cpu().numpy().copy()                       # pageable
    desc = api.QoiDesc(w, h, 4, api.QOI_SRGB)
    enc, dec = lib.qoi_encode, lib.qoi_decode
    enc.restype = ctypes.c_void_p; enc.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
    dec.restype = ctypes.c_void_p; dec.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int]
    free = ctypes.CDLL(None).free; free.argtypes = [ctypes.c_void_p]; free.restype = None
    n = ctypes.c_int(0)

    def best(fn, reps=5):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        return min(ts)
    p = enc(host_px.ctypes.data, ctypes.byref(desc), ctypes.byref(n))      # warm-up (thread context, pinned staging)
    if not p:
        return {"error": "qoi_encode failed"}
    stream_bytes = ctypes.string_at(p, n.value); free(p)
    sbuf = (ctypes.c_ubyte * len(stream_bytes)).from_buffer_copy(stream_bytes)
    # qoibench's pattern (qoibench.c:446-449, 471-475): the result is free()d inside the timed region, so the next call's malloc
    # gets the same pages back from glibc (a 31.6 MiB chunk is below the allocator's 32 MiB ceiling for its dynamic mmap
    # threshold).  "results_kept": the caller keeps every result, each call's malloc is untouched address space whose pages the
    # kernel zero-fills inside the call.
    def enc_free():
        q = enc(host_px.ctypes.data, ctypes.byref(desc), ctypes.byref(n)); free(q)
    enc_free()
    t_enc = best(enc_free)
    outs = []
    t_enc_kept = best(lambda: outs.append(enc(host_px.ctypes.data, ctypes.byref(desc), ctypes.byref(n))))
    for q in outs:
        free(q)
    d = api.QoiDesc()
    q = dec(ctypes.addressof(sbuf), len(stream_bytes), ctypes.byref(d), 4)
    ok = bool(q) and np.array_equal(np.frombuffer(ctypes.string_at(q, npx * 4), dtype=np.uint8), host_px)
    if q:
        free(q)

    def dec_free():
        q = dec(ctypes.addressof(sbuf), len(stream_bytes), ctypes.byref(d), 4); free(q)
    dec_free()
    t_dec = best(dec_free)
    outs = []
    t_dec_kept = best(lambda: outs.append(dec(ctypes.addressof(sbuf), len(stream_bytes), ctypes.byref(d), 4)))
    for q in outs:
        free(q)
    # the copies alone: pageable host <-> device, same sizes
    dpx = torch.empty(npx * 4, dtype=torch.uint8, device=dev); dst = torch.empty(len(stream_bytes), dtype=torch.uint8, device=dev)
    hpx = torch.from_numpy(host_px); hst = torch.frombuffer(bytearray(stream_bytes), dtype=torch.uint8)
    hpx2 = torch.empty_like(hpx); hst2 = torch.empty_like(hst)

    def sync(fn):
        fn(); torch.cuda.synchronize()
    sync(lambda: dpx.copy_(hpx)); sync(lambda: hst2.copy_(dst))
    c_enc = best(lambda: sync(lambda: (dpx.copy_(hpx), hst2.copy_(dst))))
    c_dec = best(lambda: sync(lambda: (dst.copy_(hst), hpx2.copy_(dpx))))
    return {"workload": f"1 x {w}x{h} RGBA frame through qoi_encode / qoi_decode on host pointers (PCIe in and out inside the call), "
                        "the result free()d inside the timed region as qoibench.c:446-449 does",
            "encode_ms": round(t_enc * 1e3, 3), "decode_ms": round(t_dec * 1e3, 3),
            "results_kept_ms": {"encode": round(t_enc_kept * 1e3, 3), "decode": round(t_dec_kept * 1e3, 3),
                                "note": "the caller keeps every result: each call's malloc is fresh address space, zero-filled by the kernel inside the call"},
            "encode_mpixels_per_s": round(npx / t_enc / 1e6, 1), "decode_mpixels_per_s": round(npx / t_dec / 1e6, 1),
            "copies_alone_ms": {"encode": round(c_enc * 1e3, 3), "decode": round(c_dec * 1e3, 3)},
            "frac_of_copies": {"encode": round(c_enc / t_enc, 3), "decode": round(c_dec / t_dec, 3)},
            "pcie_bytes": {"encode": npx * 4 + len(stream_bytes), "decode": npx * 4 + len(stream_bytes)},
            "round_trip_exact": ok}


def mixed_directory_leg(torch, api, synth, ctx, pixels, streams, decoded, lens, stream, timed) -> dict:
    """qoibench.c:491-555 walks a directory: images of different shapes and contents.  288 images, 64 distinct shapes between 48 x 48
    and 2048 x 1536, the six content classes interleaved, device-resident: ONE qoimi_encode_images (order-free placement, per-image
    table) and ONE qoimi_decode_batch; every stream hashed on the device, 72 of them against the reference encoder's."""
    from oracle import oracle_py
    lib = oracle_py.load_ref()
    checker = "reference"
    if lib is None:
        lib, checker = oracle_py.load_port(), "port"
    rng = np.random.default_rng(2026)
    kinds = ["photo", "noise", "uiflat", "constant", "photo_hard", "sprite_alpha"]
    shapes = set()
    while len(shapes) < 64:
        shapes.add((int(rng.integers(48, 2049)), int(rng.integers(48, 1537))))
    shapes = sorted(shapes)
    M = 288
    items = [(shapes[(i * 7) % len(shapes)], kinds[i % len(kinds)]) for i in range(M)]
    po, off = [], 0
    for (iw, ih), _ in items:
        po.append(off)
        off += (iw * ih * 4 + 255) // 256 * 256
    ss_m = (max(api.encode_bound(iw, ih, 4) for (iw, ih), _ in items) + 255) // 256 * 256
    ps_m = (max(iw * ih * 4 for (iw, ih), _ in items) + 255) // 256 * 256
    if off > pixels.numel() or M * ss_m > streams.numel() or M * ps_m > decoded.numel():
        return {"skipped": "the benchmark's buffers are too small for this leg (run with the default --frames)"}
    descs = [api.QoiDesc(iw, ih, 4, api.QOI_SRGB) for (iw, ih), _ in items]
    for i, ((iw, ih), kind) in enumerate(items):
        ctx.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 40000 + i, 1, iw, ih, pixels.data_ptr() + po[i], iw * ih * 4, stream)
    so = [i * ss_m for i in range(M)]
    enc = lambda: ctx.encode_images(pixels.data_ptr(), po, descs, streams.data_ptr(), so, lens.data_ptr(), stream)
    enc(); ctx.encode_status(stream)
    sizes = [int(x) for x in lens[:M].cpu().numpy()]
    dec = lambda: ctx.decode_batch(streams.data_ptr(), ss_m, sizes, descs, 4, decoded.data_ptr(), ps_m, stream)
    dec()
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.3:
        enc(); dec()
    te, td = timed(enc, 3), timed(dec, 3)
    ctx.encode_status(stream)
    rounds = ctx.decode_stats()["rounds"]
    ctx.set_profiling(True)
    enc(); dec()
    kprof = {k: round(v[0], 3) for k, v in ctx.get_profile(stream).items() if v[1] and v[0] > 0.02}
    ctx.set_profiling(False)
    ok = all(bool(torch.equal(decoded[i * ps_m:i * ps_m + items[i][0][0] * items[i][0][1] * 4], pixels[po[i]:po[i] + items[i][0][0] * items[i][0][1] * 4])) for i in range(M))
    hashes = torch.zeros(M, dtype=torch.int64, device=pixels.device)
    ctx.hash_streams(streams.data_ptr(), ss_m, lens.data_ptr(), M, hashes.data_ptr(), stream)
    torch.cuda.synchronize()
    mine = hashes.cpu().numpy().view(np.uint64)
    sample = list(range(0, M, 4))
    bad = []
    for i in sample:
        (iw, ih), kind = items[i]
        px = pixels[po[i]:po[i] + iw * ih * 4].cpu().numpy()
        if int(mine[i]) != synth.stream_hash64(lib.encode(px, iw, ih, 4)):
            bad.append(i)
    npx_total = float(sum(iw * ih for (iw, ih), _ in items))
    sbytes = float(sum(sizes))
    return {"workload": f"{M} images, {len(set(sh for sh, _ in items))} distinct shapes from 48x48 to 2048x1536, classes {'/'.join(kinds)} interleaved, device-resident: "
                        "one qoimi_encode_images + one qoimi_decode_batch (qoibench.c:491-555: a directory of images)",
            "images": M, "distinct_shapes": len(set(sh for sh, _ in items)), "mpixels": round(npx_total / 1e6, 1),
            "encode_ms": round(te * 1e3, 3), "decode_ms": round(td * 1e3, 3),
            "encode_mpixels_per_s": round(npx_total / te / 1e6, 1), "decode_mpixels_per_s": round(npx_total / td / 1e6, 1),
            "mpixels_per_s": round(npx_total / (te + td) / 1e6, 1),
            "roofline_encode_frac": round(npx_total * 4 / te / 1e9 / HBM_PEAK_GBS, 4),
            "roofline_decode_frac": round((npx_total * 4 + sbytes) / td / 1e9 / HBM_PEAK_GBS, 4),
            "stream_bytes_per_px": round(sbytes / npx_total, 4), "decode_rounds": rounds, "verified_bit_exact": bool(ok), "kernel_ms_per_call": kprof,
            "hash_check": {"checker": checker, "streams_hashed_on_device": M, "checked_against_reference": len(sample), "mismatches": bad[:8], "all_equal": not bad},
            "note": "wall clock of 3 calls each; fractions against SURVEY.md 8d's bytes and the 8 TB/s peak"}


def alternating_leg(torch, api, synth, ctx, pixels, streams, decoded, lens, stream, timed, w, h, pstride, sstride, desc) -> dict:
    """photo / uiflat / photo_hard batches of 128 4K frames in rotation on one context, no priming between them, beside the same calls
    repeated on their own class (the state in which every other leg of this file times a class)."""
    A, classes = 128, ["photo", "uiflat", "photo_hard"]
    npx = w * h
    descs = [desc] * A
    sizes = {}
    for ci, k in enumerate(classes):
        ctx.synth_frames(synth.KIND_ID[k], synth.DEFAULT_SEED, 50000 + ci * A, A, w, h, pixels.data_ptr() + ci * A * pstride, pstride, stream)
    enc = lambda ci: ctx.encode_batch(pixels.data_ptr() + ci * A * pstride, pstride, desc, A, streams.data_ptr() + ci * A * sstride, sstride, lens.data_ptr() + ci * A * 4, stream)
    dec = lambda ci: ctx.decode_batch(streams.data_ptr() + ci * A * sstride, sstride, sizes[ci], descs, 4, decoded.data_ptr() + ci * A * pstride, pstride, stream)
    first = {}
    for ci, k in enumerate(classes):
        enc(ci); ctx.encode_status(stream)
        sizes[ci] = [int(x) for x in lens[ci * A:(ci + 1) * A].cpu().numpy()]
        hs = torch.zeros(A, dtype=torch.int64, device=pixels.device)
        ctx.hash_streams(streams.data_ptr() + ci * A * sstride, sstride, lens.data_ptr() + ci * A * 4, A, hs.data_ptr(), stream)
        first[ci] = hs.cpu().numpy().copy()
        dec(ci)
    primed = {}
    for ci, k in enumerate(classes):
        for _ in range(3):
            enc(ci); dec(ci)
        primed[k] = (timed(lambda: enc(ci), 3) * 1e3, timed(lambda: dec(ci), 3) * 1e3)
    rot = {k: ([], []) for k in classes}
    same = True
    for r in range(5):                       # the first rotation is not counted (it follows the primed calls of photo_hard)
        for ci, k in enumerate(classes):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            enc(ci)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            dec(ci)
            t2 = time.perf_counter()
            if r:
                rot[k][0].append((t1 - t0) * 1e3); rot[k][1].append((t2 - t1) * 1e3)
            hs = torch.zeros(A, dtype=torch.int64, device=pixels.device)
            ctx.hash_streams(streams.data_ptr() + ci * A * sstride, sstride, lens.data_ptr() + ci * A * 4, A, hs.data_ptr(), stream)
            same = same and bool(np.array_equal(hs.cpu().numpy(), first[ci]))
    ctx.encode_status(stream)
    ok = all(bool(torch.equal(decoded[ci * A * pstride:(ci + 1) * A * pstride].view(A, pstride)[:, :npx * 4], pixels[ci * A * pstride:(ci + 1) * A * pstride].view(A, pstride)[:, :npx * 4])) for ci in range(len(classes)))
    per = {}
    for k in classes:
        e, d = float(np.mean(rot[k][0])), float(np.mean(rot[k][1]))
        per[k] = {"encode_ms": round(e, 3), "decode_ms": round(d, 3), "primed_encode_ms": round(primed[k][0], 3), "primed_decode_ms": round(primed[k][1], 3),
                  "encode_vs_primed": round(e / primed[k][0], 3), "decode_vs_primed": round(d / primed[k][1], 3)}
    return {"workload": f"batches of {A} x {w}x{h} frames, {' / '.join(classes)} in rotation on one context (4 counted rotations, every call timed with a wait before and after), "
                        "beside the same call repeated on its own class",
            "per_class": per, "streams_equal_first_encode_every_time": bool(same), "verified_bit_exact": bool(ok),
            "note": "primed_* = the call repeated on its own class (3 warm calls, then 3 timed back to back); *_vs_primed > 1.10 would mean a call-to-call heuristic costs an alternating workload more than a tenth"}


def main() -> None:
    if len(sys.argv) >= 6 and sys.argv[1] == "--cpu-worker":            # one core's share of cpu_baseline_all_cores
        print(json.dumps(cpu_baseline(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]))))
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=1024, help="4K frames resident per GPU (= per step when scaling is weak); 1024 = one GPU's shard of BASELINE configs[4]")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"], help="strong: 8192 frames in total (configs[4]), 8192/N per GPU in passes over <= --frames resident frames")
    ap.add_argument("--total-frames", type=int, default=8192, help="frames of the whole job under --scaling strong")
    ap.add_argument("--kind", default="photo", choices=["photo", "noise", "uiflat", "constant", "photo_hard", "sprite_alpha"])
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--encode-only", action="store_true", help="diagnostics: time the encoder alone (no decode, no check)")
    ap.add_argument("--no-others", action="store_true", help="skip the noise / constant / uiflat side figures")
    ap.add_argument("--no-single", action="store_true", help="skip the single-frame figure (profiling runs: keeps every launch batch-sized)")
    ap.add_argument("--no-configs", action="store_true", help="skip the BASELINE configs[2] / configs[3] side figures")
    ap.add_argument("--counter-backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend that gathers the counters (nothing else crosses ranks).  nccl = RCCL, one GPU per rank; "
                         "gloo: counters as CPU tensors, ranks may share a GPU (rank r uses device r mod device count) - how the "
                         "multi-rank path is exercised on a one-GPU box (tests/test_bench_ranks.py)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    import torch
    from qoi_amd import api, synth
    from qoi_amd import dist as qdist

    rank, world, local = qdist.env_world()
    if world != args.gpus:
        sys.exit(f"bench.py: launched with WORLD_SIZE={world} but --gpus {args.gpus}")
    n_dev = torch.cuda.device_count()
    if n_dev == 0:
        sys.exit("bench.py: no GPU visible (there is no CPU codec in this library)")
    if args.counter_backend == "nccl" and world > n_dev:
        sys.exit(f"bench.py: --gpus {world} but only {n_dev} GPU(s) visible; RCCL needs one GPU per rank "
                 f"(--counter-backend gloo lets ranks share a GPU, for testing the multi-rank path only)")
    local = local % n_dev
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    cdev = dev if args.counter_backend == "nccl" else "cpu"     # where the counter tensors live
    qdist.init(args.counter_backend, dev)          # RCCL (or gloo); only counters ever cross ranks
    ctx = api.Context(local)
    stream = torch.cuda.current_stream().cuda_stream

    w, h = args.width, args.height
    npx = w * h
    strong = args.scaling == "strong"
    if strong:
        mine = qdist.shard_range(args.total_frames, rank, world)       # this rank's frames of the job
        F = min(args.frames, len(mine))                                 # resident at a time
        passes = (len(mine) + F - 1) // F
        first_frame = mine.start
    else:
        F, passes, first_frame = args.frames, 1, rank * args.frames    # weak scaling: F distinct frames per GPU
    desc = api.QoiDesc(w, h, 4, api.QOI_SRGB)
    pstride = (npx * 4 + 255) // 256 * 256
    sstride = (api.encode_bound(w, h, 4) + 255) // 256 * 256
    pixels = torch.empty(F * pstride, dtype=torch.uint8, device=dev)
    streams = torch.empty(F * sstride, dtype=torch.uint8, device=dev)
    decoded = torch.empty(F * pstride, dtype=torch.uint8, device=dev)
    lens = torch.zeros(max(F, 1024), dtype=torch.int32, device=dev)
    ctx.synth_frames(synth.KIND_ID[args.kind], synth.DEFAULT_SEED, first_frame, F, w, h, pixels.data_ptr(), pstride, stream)
    torch.cuda.synchronize()

    # stream lengths are data-dependent; they are constant across steps, read them once
    ctx.encode_batch(pixels.data_ptr(), pstride, desc, F, streams.data_ptr(), sstride, lens.data_ptr(), stream)
    ctx.encode_status(stream)
    sizes = [int(x) for x in lens[:F].cpu().numpy()]
    descs = [desc] * F
    hash_first = torch.zeros(F, dtype=torch.int64, device=dev)        # every stream of the first encode, hashed on the device
    ctx.hash_streams(streams.data_ptr(), sstride, lens.data_ptr(), F, hash_first.data_ptr(), stream)

    # Strong scaling with more frames per rank than fit at once: every pass codes DIFFERENT frames - the resident buffer is refilled
    # with the pass's frame ids (synthetic frames are made on the device) and the stream lengths are read back after the encode,
    # as a real pipeline would.  Both happen inside the timed region and count against `value`; the refill's device time is
    # measured with events on the stream and reported beside it (`synth_ms_per_step`, `value_excluding_synth`).
    regen = strong and passes > 1
    synth_events = []
    bytes_coded = [0.0]                # stream bytes of the passes run so far (regen: lengths differ from pass to pass)

    def step():
        nonlocal sizes
        for ps in range(passes):       # strong scaling: the rank's share in passes over the resident frames
            n_now = F
            if regen:
                lo = first_frame + ps * F
                n_now = min(F, mine.stop - lo)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                ctx.synth_frames(synth.KIND_ID[args.kind], synth.DEFAULT_SEED, lo, n_now, w, h, pixels.data_ptr(), pstride, stream)
                e1.record()
                synth_events.append((e0, e1))
            ctx.encode_batch(pixels.data_ptr(), pstride, desc, n_now, streams.data_ptr(), sstride, lens.data_ptr(), stream)
            if regen:
                ctx.encode_status(stream)
                sizes = [int(x) for x in lens[:n_now].cpu().numpy()]
                bytes_coded[0] += float(sum(sizes))
            if not args.encode_only:
                ctx.decode_batch(streams.data_ptr(), sstride, sizes[:n_now], descs[:n_now], 4, decoded.data_ptr(), pstride, stream)

    def barrier():
        qdist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    ctx.set_profiling(True)
    synth_events.clear(); bytes_coded[0] = 0.0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    synth_ms = sum(a.elapsed_time(b) for a, b in synth_events)
    n_last = (mine.stop - (first_frame + (passes - 1) * F)) if regen else F        # frames of the pass that ran last (resident now)
    prof = ctx.get_profile(stream)
    ctx.set_profiling(False)
    ctx.encode_status(stream)
    ws_main = ctx.workspace_bytes()                 # what the headline batch needed (the side legs below may grow the arenas further)
    launches = args.steps * passes

    # bit-exact round trip (qoibench.c:408-417) on the whole batch, and four of its streams against the reference codec
    ok = args.encode_only or equal_batches(torch, decoded[:n_last * pstride], pixels[:n_last * pstride], n_last, pstride, npx * 4)
    dstats = ctx.decode_stats()
    refcheck = None
    if rank == 0 and not args.encode_only:
        refcheck = check_against_reference(torch, pixels, pstride, streams, sstride, sizes, w, h, sorted({0, min(1, n_last - 1), n_last // 2, n_last - 1}))
        ok = ok and refcheck["streams_byte_identical"] and refcheck["reference_decoder_round_trips"]
        # ... and byte identity as a property of the whole batch: every stream the LAST timed step wrote is hashed on the device; 64
        # frames spread over the batch must hash like the reference encoder's streams, and (one pass per step: the same frames every
        # step) all of them like the streams of the very first encode - a step that wrote other bytes anywhere would show
        hc = hash_check_against_reference(torch, ctx, pixels, pstride, streams, sstride, lens, n_last, w, h, 4, stream)
        same_as_first = None if regen else bool(np.array_equal(hc.pop("device_hashes"), hash_first[:n_last].cpu().numpy().view(np.uint64)))
        hc.pop("device_hashes", None)
        hc["every_stream_equals_first_encode"] = same_as_first
        refcheck["hash_check"] = hc
        ok = ok and hc["all_equal"] and same_as_first is not False

    def timed(fn, reps):
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / reps

    # The same decode through a record arena of 24 GiB (qoimi_set_decode_record_cap: sub-batches of whole images): what the context then
    # holds and what that costs - a report beside the headline, on its buffers, before the side legs overwrite them.
    capped = None
    if not args.encode_only and world == 1 and rank == 0 and not args.no_others and not regen and F >= 64:
        try:
            _, total_b0 = torch.cuda.mem_get_info(dev)
            ctx.set_decode_record_cap(24 << 30, True)
            dcap = lambda: ctx.decode_batch(streams.data_ptr(), sstride, sizes, descs, 4, decoded.data_ptr(), pstride, stream)
            dcap(); dcap()
            td2 = timed(dcap, 3)
            ok2 = equal_batches(torch, decoded, pixels, F, pstride, npx * 4)
            ws2 = ctx.workspace_bytes()["decode"]
            capped = {"record_cap_bytes": 24 << 30, "decode_workspace_bytes": ws2, "decode_workspace_over_stream_bytes": round(ws2 / max(1.0, float(sum(sizes))), 3),
                      "decode_ms": round(td2 * 1e3, 3), "decode_rounds": ctx.decode_stats()["rounds"], "verified_bit_exact": ok2,
                      "note": "qoimi_set_decode_record_cap(24 GiB): the headline batch decoded as sub-batches of whole images through the smaller arena (wall clock of 3 calls)"}
            ok = ok and ok2
            ctx.set_decode_record_cap(min(48 << 30, total_b0 // 6), True)       # the default again (qoimi_ctx_create)
        except Exception as e:                                                 # a report, not a gate
            capped = {"error": repr(e)}

    # BASELINE configs[1]: ONE 4K frame, encode + decode, device-resident (33 MB: served by the 256 MiB Infinity
    # Cache on repeat runs and bound by launch latency, not by HBM - reported next to the batch figure, never as it)
    single = None
    if not args.encode_only and not args.no_single and rank == 0:
        one = (ctypes.c_int * 1)(sizes[0])                 # the C caller's arguments as C arrays (int sizes[1], qoi_desc descs[1]): no per-call list conversion in the binding
        one_desc = (api.QoiDesc * 1)(desc)
        enc1 = lambda: ctx.encode_batch(pixels.data_ptr(), pstride, desc, 1, streams.data_ptr(), sstride, lens.data_ptr(), stream)
        dec1 = lambda: ctx.decode_batch(streams.data_ptr(), sstride, one, one_desc, 4, decoded.data_ptr(), pstride, stream)
        # (the reference checks above keep the host busy and the GPU idle for seconds: its clocks have dropped, and 20 launches of 40 us
        # do not bring them back - 0.3 s of the same calls first, then the timed ones)
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.3:
            enc1(); dec1()
        med3 = lambda fn: sorted(timed(fn, 50) for _ in range(3))[1]      # (median of three blocks of 50 calls: one block now and then catches a hiccup of 25 %)
        dt = med3(lambda: (enc1(), dec1()))
        dt_e = med3(enc1)
        dt_d = med3(dec1)
        # the same encode with its units handed out by workgroup index (qoimi_set_encode_small_call_order: what the drop-in qoi_encode
        # takes; a caller of the device API must then ask qoimi_encode_status before reading the streams)
        ctx.set_encode_small_call_order(True)
        for _ in range(20):
            enc1()
        dt_e_idx = med3(enc1)
        ctx.encode_status(stream)
        ctx.set_encode_small_call_order(False)
        enc1(); ctx.encode_status(stream)
        single = {"workload": f"1 x {w}x{h} RGBA frame, encode + decode, device-resident, wall clock incl. launches (Infinity-Cache resident on repeat runs); every figure the median of three blocks of 50 calls",
                  "ms": round(dt * 1e3, 4), "mpixels_per_s": round(npx / dt / 1e6, 1),
                  "encode_ms": round(dt_e * 1e3, 4), "decode_ms": round(dt_d * 1e3, 4),
                  "encode_ms_units_by_workgroup_index": round(dt_e_idx * 1e3, 4),
                  "placement_note": "encode_ms: the default, units by ticket (start order: safe when launches of several streams share the device); "
                                    "by workgroup index: opt-in (qoimi_set_encode_small_call_order), qoimi_encode_status then is mandatory",
                  "roofline": {"bound": "hbm", "kernel": "whole single-frame encode (all launches)", "achieved": round(npx * 4 / dt_e / 1e9, 1), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": round(npx * 4 / dt_e / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                               "algorithmic_bytes_per_launch": npx * 4, "ms_per_launch": round(dt_e * 1e3, 4),
                               "target": "BASELINE.md: 50 % of the roofline = 8.3 us per 4K frame"}}

    # The reference's own entry points (qoi.h:278/289) on HOST pointers: pixels and stream cross PCIe in both directions inside the
    # call.  Reported beside the copies alone (the same bytes, the same pageable buffers, torch copy_), never as `value`.
    dropin = None
    if not args.encode_only and not args.no_single and rank == 0 and world == 1:
        dropin = dropin_path(torch, api, ctx, pixels, w, h, dev)

    # next to the headline content always report `noise` (5 B/px written: most stream traffic) and
    # `constant` (longest runs) - same batch, 3 timed steps each, rank 0 of a single-GPU run only
    other = None
    if world == 1 and not args.encode_only and not args.no_others:
        other = {}
        for kind in ("noise", "constant", "uiflat", "photo_hard", "sprite_alpha", "photo"):
            if kind == args.kind:
                continue
            ctx.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 0, F, w, h, pixels.data_ptr(), pstride, stream)
            ctx.encode_batch(pixels.data_ptr(), pstride, desc, F, streams.data_ptr(), sstride, lens.data_ptr(), stream)
            ctx.encode_status(stream)
            ksizes = [int(x) for x in lens[:F].cpu().numpy()]
            ctx.decode_batch(streams.data_ptr(), sstride, ksizes, descs, 4, decoded.data_ptr(), pstride, stream)   # warm-up
            enc_k = lambda: ctx.encode_batch(pixels.data_ptr(), pstride, desc, F, streams.data_ptr(), sstride, lens.data_ptr(), stream)
            dec_k = lambda: ctx.decode_batch(streams.data_ptr(), sstride, ksizes, descs, 4, decoded.data_ptr(), pstride, stream)
            # the reference checks of the class before kept the host busy for seconds and the device idle: a class whose calls take a few
            # milliseconds (constant: 6 + 8) would be timed while the clocks are still on their way up (one reduced run read its encode at
            # 10.7-14.6 ms, profiles/r05_s32_*) - 0.3 s of the class's own calls first, as for the single frame
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < 0.3:
                enc_k(); dec_k()
                torch.cuda.synchronize()
            te, td = timed(enc_k, 3), timed(dec_k, 3)
            dt = te + td
            kok = equal_batches(torch, decoded, pixels, F, pstride, npx * 4)
            # two frames of every content class against the REFERENCE codec (a valid, round-tripping but longer stream - round 1's
            # uiflat error - is only visible in a byte comparison) and 64 frames spread over the batch by hash
            kchk = check_against_reference(torch, pixels, pstride, streams, sstride, ksizes, w, h, sorted({0, F - 1})) if rank == 0 else None
            if kchk is not None:
                khc = hash_check_against_reference(torch, ctx, pixels, pstride, streams, sstride, lens, F, w, h, 4, stream)
                khc.pop("device_hashes", None)
                kchk["hash_check"] = khc
            kok = kok and (kchk is None or (kchk["streams_byte_identical"] and kchk["reference_decoder_round_trips"] and kchk["hash_check"]["all_equal"]))
            kbytes = float(sum(ksizes))
            other[kind] = {"mpixels_per_s": round(F * npx / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 3),
                           "encode_ms": round(te * 1e3, 3), "decode_ms": round(td * 1e3, 3),
                           "roofline_encode_frac": round(F * npx * 4 / te / 1e9 / HBM_PEAK_GBS, 4),
                           "roofline_decode_frac": round((F * npx * 4 + kbytes) / td / 1e9 / HBM_PEAK_GBS, 4),
                           "stream_bytes_per_px": round(kbytes / (F * npx), 4), "decode_rounds": ctx.decode_stats()["rounds"],
                           "verified_bit_exact": kok, "reference_check": kchk,
                           "note": "wall clock of 3 whole qoimi_encode_batch / qoimi_decode_batch calls each; fractions against SURVEY.md 8d's bytes "
                                   "(encode: 4 B read per pixel; decode: stream bytes + 4 B written per pixel) and the 8 TB/s peak"}

    # the same photographs as 3-channel input and output (qoi.h:406-413, 580-586: channels = 3), a quarter of the batch
    rgb = None
    if world == 1 and rank == 0 and not args.encode_only and not args.no_others:
        F3 = max(1, min(F, 256))
        ps3 = (npx * 3 + 255) // 256 * 256
        d3 = api.QoiDesc(w, h, 3, api.QOI_SRGB)
        ctx.synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 0, F3, w, h, pixels.data_ptr(), pstride, stream)
        torch.cuda.synchronize()
        src3 = decoded[:F3 * ps3].view(F3, ps3)
        for lo in range(0, F3, 16):                                    # r,g,b of every pixel, tightly packed (no batch-sized temporary)
            hi3 = min(F3, lo + 16)
            src3[lo:hi3, :npx * 3] = pixels[lo * pstride:hi3 * pstride].view(hi3 - lo, pstride)[:, :npx * 4].view(hi3 - lo, npx, 4)[:, :, :3].reshape(hi3 - lo, npx * 3)
        out3 = pixels[:F3 * ps3]
        ctx.encode_batch(decoded.data_ptr(), ps3, d3, F3, streams.data_ptr(), sstride, lens.data_ptr(), stream)
        ctx.encode_status(stream)
        s3 = [int(x) for x in lens[:F3].cpu().numpy()]
        ctx.decode_batch(streams.data_ptr(), sstride, s3, [d3] * F3, 3, out3.data_ptr(), ps3, stream)                  # warm-up
        e3 = lambda: ctx.encode_batch(decoded.data_ptr(), ps3, d3, F3, streams.data_ptr(), sstride, lens.data_ptr(), stream)
        x3 = lambda: ctx.decode_batch(streams.data_ptr(), sstride, s3, [d3] * F3, 3, out3.data_ptr(), ps3, stream)
        te, td = timed(e3, 3), timed(x3, 3)
        ok3 = equal_batches(torch, out3, decoded[:F3 * ps3], F3, ps3, npx * 3)
        from oracle import oracle_py
        lib3 = oracle_py.load_ref() or oracle_py.load_port()
        ident3 = True
        for fr in sorted({0, F3 - 1}):
            px3 = decoded[fr * ps3:fr * ps3 + npx * 3].cpu().numpy()
            ident3 = ident3 and streams[fr * sstride:fr * sstride + s3[fr]].cpu().numpy().tobytes() == lib3.encode(px3, w, h, 3)
        rgb = {"workload": f"{F3} x {w}x{h} photo frames with 3 channels in and out, encode + decode, HBM-resident",
               "mpixels_per_s": round(F3 * npx / (te + td) / 1e6, 1), "encode_ms": round(te * 1e3, 3), "decode_ms": round(td * 1e3, 3),
               "encode_mpixels_per_s": round(F3 * npx / te / 1e6, 1), "decode_mpixels_per_s": round(F3 * npx / td / 1e6, 1),
               "stream_bytes_per_px": round(sum(s3) / (F3 * npx), 4), "verified_bit_exact": bool(ok3), "streams_byte_identical_to_reference": bool(ident3)}

    # BASELINE configs[2]: 1024 x 1920x1080 RGBA, encode only (the HBM-bound roofline run) and configs[3]: one 16384 x 16384
    # image, encode + decode - in the buffers of the main batch where they fit
    cfg2 = cfg3 = None
    if world == 1 and rank == 0 and not args.no_configs and not args.encode_only:
        w2, h2, F2 = 1920, 1080, 1024
        n2 = w2 * h2
        ps2, ss2 = (n2 * 4 + 255) // 256 * 256, (api.encode_bound(w2, h2, 4) + 255) // 256 * 256
        if F2 * ps2 <= pixels.numel() and F2 * ss2 <= streams.numel():
            d2 = api.QoiDesc(w2, h2, 4, api.QOI_SRGB)
            ctx.synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 20000, F2, w2, h2, pixels.data_ptr(), ps2, stream)
            e2 = lambda: ctx.encode_batch(pixels.data_ptr(), ps2, d2, F2, streams.data_ptr(), ss2, lens.data_ptr(), stream)
            e2(); e2()
            ctx.set_profiling(True)
            dt = timed(e2, 10)
            p2 = ctx.get_profile(stream)
            ctx.set_profiling(False)
            ctx.encode_status(stream)
            s2 = [int(x) for x in lens[:F2].cpu().numpy()]
            chk = check_against_reference(torch, pixels, ps2, streams, ss2, s2, w2, h2, [0, F2 - 1])
            hc2 = hash_check_against_reference(torch, ctx, pixels, ps2, streams, ss2, lens, F2, w2, h2, 4, stream)
            hc2.pop("device_hashes", None)
            chk["hash_check"] = hc2
            slabs_ms = sum(p2[k][0] for k in ("enc_slabs", "enc_slabs_generic", "enc_slab_summary", "enc_scan_groups", "enc_scan_images") if k in p2) / 10
            tot_ms = p2["encode_total"][0] / 10 if p2.get("encode_total", (0, 0))[1] else dt * 1e3
            cfg2 = {"workload": f"BASELINE configs[2]: {F2} x {w2}x{h2} RGBA frames (photo), encode only, HBM-resident ({F2 * n2 * 4 / 1e9:.2f} GB of pixels per launch)",
                    "ms_per_step": round(dt * 1e3, 4), "mpixels_per_s": round(F2 * n2 / dt / 1e6, 1),
                    "stream_bytes_per_px": round(sum(s2) / (F2 * n2), 4), "reference_check": chk,
                    "roofline": {"bound": "hbm", "kernel": "enc_sets (+ entry-state passes)", "achieved": round(F2 * n2 * 4 / (slabs_ms * 1e-3) / 1e9, 1),
                                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(F2 * n2 * 4 / (slabs_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                 "traffic": None, "algorithmic_bytes_per_launch": F2 * n2 * 4, "ms_per_launch": round(slabs_ms, 4)},
                    "roofline_encode_total": {"bound": "hbm", "kernel": "whole qoimi_encode_batch (all kernels)", "achieved": round(F2 * n2 * 4 / (tot_ms * 1e-3) / 1e9, 1),
                                              "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(F2 * n2 * 4 / (tot_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                              "algorithmic_bytes_per_launch": F2 * n2 * 4, "ms_per_launch": round(tot_ms, 4)}}
        w3 = h3 = 16384
        n3 = w3 * h3
        ps3, ss3 = n3 * 4, (api.encode_bound(w3, h3, 4) + 255) // 256 * 256
        if ps3 <= pixels.numel() and ss3 <= streams.numel():
            d3 = api.QoiDesc(w3, h3, 4, api.QOI_SRGB)
            ctx.synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 30000, 1, w3, h3, pixels.data_ptr(), ps3, stream)
            e3 = lambda: ctx.encode_batch(pixels.data_ptr(), ps3, d3, 1, streams.data_ptr(), ss3, lens.data_ptr(), stream)
            e3()
            ctx.encode_status(stream)
            s3 = [int(lens[0].item())]
            g3 = lambda: ctx.decode_batch(streams.data_ptr(), ss3, s3, [d3], 4, decoded.data_ptr(), ps3, stream)
            g3()
            t_warm = time.perf_counter()
            while time.perf_counter() - t_warm < 0.3:              # (clocks: see other_content)
                e3(); g3()
                torch.cuda.synchronize()
            dte, dtd = timed(e3, 5), timed(g3, 5)
            ok3 = bool(torch.equal(decoded[:ps3], pixels[:ps3]))
            cfg3 = {"workload": f"BASELINE configs[3]: one {w3}x{h3} RGBA image (photo, {n3 / 1e6:.0f} Mpx), encode + decode, device-resident, wall clock incl. launches",
                    "encode_ms": round(dte * 1e3, 3), "decode_ms": round(dtd * 1e3, 3), "mpixels_per_s": round(n3 / (dte + dtd) / 1e6, 1),
                    "encode_mpixels_per_s": round(n3 / dte / 1e6, 1), "decode_mpixels_per_s": round(n3 / dtd / 1e6, 1),
                    "stream_bytes_per_px": round(s3[0] / n3, 4), "decode_rounds": ctx.decode_stats()["rounds"], "verified_bit_exact": ok3,
                    "roofline_encode_total": {"bound": "hbm", "kernel": "whole qoimi_encode_batch (all kernels, wall clock)", "achieved": round(n3 * 4 / dte / 1e9, 1), "peak": HBM_PEAK_GBS,
                                              "unit": "GB/s", "frac": round(n3 * 4 / dte / 1e9 / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": n3 * 4, "ms_per_launch": round(dte * 1e3, 3)}}

    # What the reference's own harness walks (qoibench.c:491-555): a DIRECTORY of images of different shapes and contents, here one
    # qoimi_encode_images + one qoimi_decode_batch over 288 images of 64 shapes, the six content classes interleaved.
    mixed = None
    if world == 1 and rank == 0 and not args.no_others and not args.encode_only:
        try:
            mixed = mixed_directory_leg(torch, api, synth, ctx, pixels, streams, decoded, lens, stream, timed)
        except Exception as e:                                                  # a report, not a gate
            mixed = {"error": repr(e)}
    # ... and batches of different content in rotation on ONE context, nothing primed: what the call-to-call heuristics (set size by the
    # previous batch's bytes per pixel, pass selection by the previous batch's flagged images) cost a workload that alternates
    alternating = None
    if world == 1 and rank == 0 and not args.no_others and not args.encode_only and F >= 384 and (w, h) == (3840, 2160):
        try:
            alternating = alternating_leg(torch, api, synth, ctx, pixels, streams, decoded, lens, stream, timed, w, h, pstride, sstride, desc)
        except Exception as e:
            alternating = {"error": repr(e)}

    # RCCL: counters only (max elapsed; summed pixels / stream bytes / verified ranks)
    # synthetic frame ids this rank coded in a step: its whole share when every pass refills the buffer, else the resident frames
    # (passes > 1 always refills; with one pass the resident frames ARE the share)
    my_frames = range(first_frame, first_frame + (len(mine) if strong else F))
    px_coded = float(len(my_frames) * npx * args.steps) if regen else float(F * npx * launches)
    st_coded = bytes_coded[0] if regen else float(sum(sizes)) * launches
    elapsed_mine = elapsed
    elapsed_min = qdist.reduce_min(elapsed_mine, cdev)               # the fastest rank (skew between ranks = max - min)
    free_mine, total_mine = torch.cuda.mem_get_info(dev)
    peak_max, (peak_sum,) = qdist.reduce_counters(float(total_mine - free_mine), [float(total_mine - free_mine)], cdev)   # every rank's device: the largest, and the sum
    elapsed, (total_px, total_stream_bytes, n_ok, frames_coded, frame_id_sum, synth_ms_max) = qdist.reduce_counters(
        elapsed, [px_coded, st_coded, float(ok), float(len(my_frames)), float(sum(my_frames)), float(synth_ms)], cdev)

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        value = total_px / elapsed / 1e6
        # whole calls on the launch stream
        enc_ms = prof["encode_total"][0] if prof.get("encode_total", (0, 0))[1] else sum(prof[k][0] for k in prof if k.startswith("enc_"))
        dec_ms = prof["decode_total"][0] if prof.get("decode_total", (0, 0))[1] else sum(prof[k][0] for k in prof if k.startswith("dec_"))
        # The encode kernel of the roofline: enc_sets (timer tag "enc_slabs") plus the entry-state passes that run before its second launch
        # for images the first launch could not finish on its own (flat content) - every kernel that reads pixels.
        slabs_calls = prof["enc_slabs"][1]
        slabs_ms = sum(prof[k][0] for k in ("enc_slabs", "enc_slabs_generic", "enc_slab_summary", "enc_scan_groups", "enc_scan_images") if k in prof)
        per_launch_ms = slabs_ms / max(1, slabs_calls)
        frames_per_launch = len(my_frames) / passes if regen else F
        alg_bytes = frames_per_launch * npx * 4           # 4 B read per pixel
        stream_bytes = total_stream_bytes / max(1, launches) / world          # per launch of this rank
        gbs = lambda nbytes, ms: nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        achieved = gbs(alg_bytes, per_launch_ms)
        # the kernel with the largest share of the step.  A decode call may run as several sub-batches (record arena cap):
        # bytes and duration are both per LAUNCH (the call's bytes / its launches of this kernel).
        seg_calls = prof["dec_segments"][1] if "dec_segments" in prof else 0
        seg_ms = prof["dec_segments"][0] / max(1, seg_calls) if seg_calls else 0.0
        seg_per_call = seg_calls / max(1, launches) if seg_calls else 1.0
        dec_bytes = alg_bytes + stream_bytes               # stream read + 4 B written per pixel (whole decode)
        enc_tot_ms, dec_tot_ms = enc_ms / max(1, launches), dec_ms / max(1, launches)

        # PMC traffic of enc_sets from the committed passes of this command (same frames; per-frame traffic scales with the batch)
        traffic = traffic_src = None
        if args.kind == "photo" and (w, h) == (3840, 2160):
            try:
                f_prof = float(json.load(open(TRAFFIC_FILE)).get("frames", F))
            except (OSError, ValueError):
                f_prof = float(F)
            traffic, traffic_src = measured_traffic("qoimi::enc_sets<4, 1, 1, false, false>", frames_per_launch / f_prof)
        roof = lambda kernel, nbytes, ms, **kw: dict({"bound": "hbm", "kernel": kernel, "achieved": round(gbs(nbytes, ms), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                      "frac": round(gbs(nbytes, ms) / HBM_PEAK_GBS, 4), "algorithmic_bytes_per_launch": int(nbytes), "ms_per_launch": round(ms, 4)}, **kw)
        out = {
            "metric": "Mpixels/s encode+decode, 4K RGBA", "value": round(value, 1), "unit": "Mpixels/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_step, 4),
            "ms_per_step_ranks": {"min": round(elapsed_min / args.steps * 1e3, 4), "max": round(ms_step, 4), "rank0": round(elapsed_mine / args.steps * 1e3, 4),
                                  "note": "wall clock of the timed region per rank / steps; ms_per_step is the slowest rank's"},
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": "u8", "data": "synthetic",
            "config": {"workload": (f"batch of {F} x {w}x{h} RGBA frames per GPU per step = one GPU's shard of BASELINE configs[4] (8192 frames over 8 GPUs)"
                                    if not strong else
                                    f"BASELINE configs[4] strong scaling: {args.total_frames} x {w}x{h} RGBA frames in total, {args.total_frames // world} per GPU per step "
                                    f"in {passes} pass(es) over {F} resident frames") +
                                   f", encode + decode, content={args.kind}, HBM-resident, bit-exact round trip verified",
                       "frames_per_gpu": len(my_frames), "frames_resident": F, "passes_per_step": passes, "width": w, "height": h, "content": args.kind,
                       "stream_bytes_per_px": round(total_stream_bytes / total_px, 4), "parallelism": f"frames sharded x{world}"},
            "verified_bit_exact": n_ok == world, "reference_check": refcheck,
            "distinct_frames_per_step": "every pass refills the resident buffer with its own frame ids (refill and stream-length read-back inside the timed region)" if regen
                                        else "the resident frames (one pass per step)",
            "counter_backend": args.counter_backend, "frames_coded_all_ranks": int(frames_coded), "frame_id_sum_all_ranks": int(frame_id_sum),
            "encode_mpps_kernels": round(F * npx * launches / (enc_ms * 1e3), 1) if enc_ms else None,
            "decode_mpps_kernels": round(F * npx * launches / (dec_ms * 1e3), 1) if dec_ms else None,
            "decode_rounds": dstats["rounds"], "decode_redo_segments": dstats["redo_segments"], "decode_sync_fallback_segments": dstats.get("sync_fallback_segments"),
            "kernel_ms_per_step": {k: round(v[0] / args.steps, 4) for k, v in prof.items() if v[1]},
            "roofline": roof("enc_sets (+ entry-state passes)", alg_bytes, per_launch_ms, traffic=traffic, traffic_source=traffic_src,
                             note="the kernel of the north star's 4K-encode roofline target; the kernel with the largest share of the step is in roofline_dominant; "
                                  "ms_per_launch / frac: HIP events on the launch stream in THIS run; rocprof_*: the latest committed rocprofv3 summary of this command "
                                  "(another session, another box): the two clocks bracket the figure, 0.326-0.341 over round 5's sessions"),
        }
        if args.kind == "photo" and (w, h) == (3840, 2160) and frames_per_launch == 1024:
            rp_ms, rp_src = rocprof_average_ms("enc_sets<4, 1, 1, false, false>")
            if rp_ms:
                out["roofline"].update({"rocprof_ms_per_launch": round(rp_ms, 4), "rocprof_frac": round(gbs(alg_bytes, rp_ms) / HBM_PEAK_GBS, 4), "rocprof_source": rp_src})
        if not args.encode_only:
            # PMC traffic of the kernel's largest launch (a sub-batch of the decode call), scaled to the average launch by segments
            seg_traffic = seg_src = None
            try:
                kdoc = json.load(open(TRAFFIC_FILE))
                for name, k in kdoc["kernels"].items():
                    if name in ("qoimi::dec_segments_rec<4, false>", "qoimi::dec_segments_rec<4>") and "FETCH_SIZE_KB" in k and "WRITE_SIZE_KB" in k and k.get("grid") and dstats.get("segments"):
                        sc = (dstats["segments"] / seg_per_call) / float(k["grid"])
                        seg_traffic = (k["FETCH_SIZE_KB"] * kdoc["read_correction"] + k["WRITE_SIZE_KB"]) * 1024.0 * sc
                        seg_src = (f"profiles/{os.path.basename(TRAFFIC_FILE)}: {name}, launch of {k['grid']} segment lanes (2 x FETCH_SIZE + WRITE_SIZE), "
                                   f"x {sc:.3f} = segments of an average launch of this run")
            except (OSError, ValueError, KeyError):
                pass
            out["roofline_dominant"] = roof("dec_segments_rec", dec_bytes / seg_per_call, seg_ms, traffic=seg_traffic, traffic_source=seg_src, launches_per_decode_call=round(seg_per_call, 2),
                                            note="largest share of the step; algorithmic bytes as SURVEY.md 8d defines them for decode (stream bytes + 4 B written per pixel) - "
                                                 "the kernel itself reads one 4-byte chunk record per chunk instead of the stream (DESIGN.md section 4)")
            out["roofline_decode_total"] = roof("whole qoimi_decode_batch (all kernels)", dec_bytes, dec_tot_ms, note="SURVEY.md 8d: stream bytes read + 4 B written per pixel")
        out["roofline_encode_total"] = roof("whole qoimi_encode_batch (all kernels)", alg_bytes, enc_tot_ms, note="SURVEY.md 8d: 4 B read per pixel")
        if not args.encode_only:
            # the whole step against SURVEY.md 8d's bytes for encode + decode (pixels read, stream written and read back, pixels written)
            out["roofline_step"] = roof("whole step: qoimi_encode_batch + qoimi_decode_batch (wall clock)", 2 * alg_bytes + 2 * stream_bytes, ms_step / max(1, passes),
                                        note="SURVEY.md 8d: 4 B read per pixel + stream bytes written (encode), stream bytes read + 4 B written per pixel (decode)")
        if regen:
            sm = synth_ms_max / world                         # mean over the ranks (they refill the same number of frames)
            out["synth_ms_per_step"] = round(sm / args.steps, 3)
            out["value_excluding_synth"] = round(total_px / max(1e-9, elapsed - sm * 1e-3) / 1e6, 1)
        # device memory: what the context's arenas hold now (they only grow) and what the process has taken from the device
        ws = ctx.workspace_bytes()
        free_b, total_b = torch.cuda.mem_get_info(dev)
        out["device_memory"] = {"encode_workspace_bytes": ws["encode"], "decode_workspace_bytes": ws["decode"], "dropin_staging_bytes": ws["staging"],
                                "headline_batch": {"encode_workspace_bytes": ws_main["encode"], "decode_workspace_bytes": ws_main["decode"], "stream_bytes": int(stream_bytes),
                                                   "decode_workspace_over_stream_bytes": round(ws_main["decode"] / max(1.0, stream_bytes), 3) if not args.encode_only else None,
                                                   "note": "decode: 4 bytes of chunk records per stream byte reserved (one record per byte is the worst case) + ~0.27 of per-segment state; a caller short of device memory caps the record arena (qoimi_set_decode_record_cap / QOIMI_DEC_REC_CAP_MB) and the call runs as sub-batches of whole images: see record_cap_24g",
                                                   "record_cap_24g": capped},
                                "bench_buffers_bytes": int(pixels.numel() + streams.numel() + decoded.numel()),
                                "peak_device_bytes": int(total_b - free_b), "device_total_bytes": int(total_b),
                                "peak_device_bytes_per_rank": {"max": int(peak_max), "sum_over_ranks": int(peak_sum), "ranks": world,
                                                               "note": "device total - free on each rank's own GPU when its timed region ended (before rank 0's side legs); every rank must fit ITS 288 GB"},
                                "note": "peak_device_bytes = device total - free at the end of the run (all processes on the device; the library's arenas and torch's caching allocator only grow)"}
        out["scaling_note"] = ("single GPU" if world == 1 else f"{world} ranks") + "; no multi-GPU scaling curve has been measured for this repository (gpurun exposes one GPU) - the driver computes efficiency from its own per-N runs"
        if single:
            out["single_frame"] = single
        if dropin is not None:
            out["dropin_host_pointers"] = dropin
        if cfg2:
            out["encode_1080p_batch"] = cfg2
        if cfg3:
            out["single_16k"] = cfg3
        if mixed:
            out["mixed_directory"] = mixed
        if alternating:
            out["alternating"] = alternating
        if other:
            out["other_content"] = other
        if rgb:
            out["rgb_input"] = rgb
        if args.encode_only:
            out["config"]["workload"] += " [ENCODE ONLY - diagnostic run, not the benchmark]"
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # The reference CPU path beside the GPU figure, for EVERY world size: rank 0 times it after the final barrier (the other
        # ranks are through; the GPUs are idle), so the line the driver parses at N = 2, 4, 8 carries the same objects as at N = 1.
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args.kind, w, h, args.cpu_seconds)
            out["cpu_baseline_all_cores"] = cpu_baseline_all_cores(args.kind, w, h, min(args.cpu_seconds, 6.0))
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
