#!/usr/bin/env python
"""Stress of the placement forms that WAIT on other sets (tree, look-back) under contention: T host threads, each with its own
context and HIP stream, encode one device-resident 4K frame per call at the same time - thousands of workgroups of several
kernels compete for the chip's ~1500 workgroup slots, so every kernel's workgroups start late and interleaved.  Every stream must
equal the reference's, qoimi_encode_status must never report a tripped spin bound, decodes interleave.

    python tests/stress_threads.py --threads 8 --calls 40            # needs an MI355X
"""
import argparse
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--calls", type=int, default=40)
    ap.add_argument("--width", type=int, default=3840)
    ap.add_argument("--height", type=int, default=2160)
    ap.add_argument("--batch-frames", type=int, default=0, help="one MORE thread encodes + decodes batches of this many frames (its own context and stream) while the others run their single frames")
    ap.add_argument("--default-placement", action="store_true", help="every single-frame context takes the library's default placement (tickets) instead of the forced mix")
    a = ap.parse_args()
    os.environ["QOIMI_TUNING"] = "1"                                  # (the placement forms below are selected by knobs)
    import torch
    from oracle import oracle_py
    from qoi_amd import api, synth
    ref = oracle_py.load_ref() or oracle_py.load_port()
    w, h = a.width, a.height
    npx = w * h
    kinds = ["photo", "uiflat", "noise", "photo", "constant", "photo", "noise", "uiflat"]
    want = {}
    for k in set(kinds):
        want[k] = np.frombuffer(ref.encode(synth.frame_rgba(k, w, h, 3), w, h, 4), dtype=np.uint8)
    errors = []
    start = threading.Barrier(a.threads + (1 if a.batch_frames else 0))
    stop_batches = threading.Event()
    forms = ["", "2", "1", "", "2", "0", "", "1"]
    if a.default_placement:
        forms = [""] * 8
    batch_note = []

    def batch_work():
        """the fourth context of round 5's review: whole batches on a stream of their own while the single frames run"""
        try:
            c = api.Context(0)
            st = torch.cuda.Stream()
            F = a.batch_frames
            with torch.cuda.stream(st):
                ps = (npx * 4 + 255) // 256 * 256
                ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
                px = torch.empty(F * ps, dtype=torch.uint8, device="cuda")
                out = torch.empty(F * ps, dtype=torch.uint8, device="cuda")
                sb = torch.empty(F * ss, dtype=torch.uint8, device="cuda")
                ln = torch.zeros(F, dtype=torch.int32, device="cuda")
                hs = torch.zeros(F, dtype=torch.int64, device="cuda")
                c.synth_frames(synth.KIND_ID["photo"], synth.DEFAULT_SEED, 1000, F, w, h, px.data_ptr(), ps, st.cuda_stream)
                desc = api.QoiDesc(w, h, 4, 0)
                st.synchronize()
                first = None
                start.wait()
                n_calls = 0
                while not stop_batches.is_set() or n_calls < 2:
                    c.encode_batch(px.data_ptr(), ps, desc, F, sb.data_ptr(), ss, ln.data_ptr(), st.cuda_stream)
                    c.encode_status(st.cuda_stream)
                    c.hash_streams(sb.data_ptr(), ss, ln.data_ptr(), F, hs.data_ptr(), st.cuda_stream)
                    sizes = [int(x) for x in ln.cpu().numpy()]
                    h_now = hs.cpu().numpy().copy()
                    if first is None:
                        first = h_now
                        for f in (0, F // 2, F - 1):                   # three streams of the first batch against the reference encoder
                            host = synth.frame_rgba("photo", w, h, 1000 + f)
                            if sb[f * ss:f * ss + sizes[f]].cpu().numpy().tobytes() != ref.encode(host, w, h, 4):
                                errors.append(("batch", n_calls, f, "encode"))
                    elif not np.array_equal(first, h_now):
                        errors.append(("batch", n_calls, "streams differ from the first batch's"))
                    c.decode_batch(sb.data_ptr(), ss, sizes, [desc] * F, 4, out.data_ptr(), ps, st.cuda_stream)
                    if not torch.equal(out.view(F, ps)[:, :npx * 4], px.view(F, ps)[:, :npx * 4]):
                        errors.append(("batch", n_calls, "decode"))
                    n_calls += 1
                if c.encode_retries():
                    errors.append(("batch", "retries", c.encode_retries()))
                batch_note.append(f"; beside them one context round-tripped {n_calls} batches of {F} frames")
        except Exception as e:                                                       # noqa: BLE001
            errors.append(("batch", repr(e)))

    def work(t):
        try:
            kind = kinds[t % len(kinds)]
            os.environ["QOIMI_ENC_LOOKBACK"] = forms[t % len(forms)]       # (read at context creation; the threads create theirs one after the other below)
            c = ctxs[t]
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                ps = npx * 4
                ss = (api.encode_bound(w, h, 4) + 255) // 256 * 256
                px = torch.empty(ps, dtype=torch.uint8, device="cuda")
                out = torch.empty(ps, dtype=torch.uint8, device="cuda")
                sb = torch.zeros(ss, dtype=torch.uint8, device="cuda")
                ln = torch.zeros(4, dtype=torch.int32, device="cuda")
                c.synth_frames(synth.KIND_ID[kind], synth.DEFAULT_SEED, 3, 1, w, h, px.data_ptr(), ps, st.cuda_stream)
                desc = api.QoiDesc(w, h, 4, 0)
                st.synchronize()
                start.wait()
                for it in range(a.calls):
                    c.encode_batch(px.data_ptr(), ps, desc, 1, sb.data_ptr(), ss, ln.data_ptr(), st.cuda_stream)
                    c.encode_status(st.cuda_stream)                                  # raises on a tripped spin bound
                    n = int(ln[0].item())
                    if n != len(want[kind]) or not np.array_equal(sb[:n].cpu().numpy(), want[kind]):
                        errors.append((t, it, kind, "encode", n, len(want[kind])))
                    if it % 4 == 3:
                        c.decode_batch(sb.data_ptr(), ss, [n], [desc], 4, out.data_ptr(), ps, st.cuda_stream)
                        if not torch.equal(out, px):
                            errors.append((t, it, kind, "decode"))
                if c.encode_retries():
                    errors.append((t, "placement waits gave up", c.encode_retries()))
        except Exception as e:                                                       # noqa: BLE001
            errors.append((t, repr(e)))

    ctxs = []
    for t in range(a.threads):
        os.environ["QOIMI_ENC_LOOKBACK"] = forms[t % len(forms)]
        if not forms[t % len(forms)]:
            os.environ.pop("QOIMI_ENC_LOOKBACK")
        ctxs.append(api.Context(0))
    t0 = time.time()
    th = [threading.Thread(target=work, args=(t,)) for t in range(a.threads)]
    bt = threading.Thread(target=batch_work) if a.batch_frames else None
    for x in th + ([bt] if bt else []):
        x.start()
    for x in th:
        x.join()
    stop_batches.set()
    if bt:
        bt.join()
    print(f"stress_threads: {a.threads} threads x {a.calls} single-frame encodes of {w}x{h} ({', '.join(kinds[:a.threads])}; placement {forms[:a.threads]}) "
          f"at the same time, a decode every fourth call{''.join(batch_note)}: {'all byte-identical to the ' + ref.kind + ' encoder, no placement wait gave up' if not errors else errors[:6]}; {time.time() - t0:.1f} s")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
