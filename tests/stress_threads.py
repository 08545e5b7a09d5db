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
    a = ap.parse_args()
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
    start = threading.Barrier(a.threads)
    forms = ["", "2", "1", "", "2", "0", "", "1"]

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
    for x in th:
        x.start()
    for x in th:
        x.join()
    print(f"stress_threads: {a.threads} threads x {a.calls} single-frame encodes of {w}x{h} ({', '.join(kinds[:a.threads])}; placement {forms[:a.threads]}) "
          f"at the same time, a decode every fourth call: {'all byte-identical to the ' + ref.kind + ' encoder, no spin bound tripped' if not errors else errors[:6]}; {time.time() - t0:.1f} s")
    return 1 if errors else 0


if __name__ == "__main__":
    sys.exit(main())
