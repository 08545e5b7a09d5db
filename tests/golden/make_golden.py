"""Generate tests/golden/qoi_golden.npz from the UNMODIFIED reference.

Run in the authoring container (needs /root/reference):
    python tests/golden/make_golden.py
It builds oracle/_ref/libqoiref.so (oracle/Makefile: gcc on /root/reference/qoi.h with
renamed symbols), runs the reference's qoi_encode / qoi_decode on every case of
tests/cases.py and stores the outputs.  The .npz is committed; the reference is not
needed to run the tests.
"""
import os
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle_py as O  # noqa: E402
import cases  # noqa: E402


def main():
    O.build()
    ref = O.load_ref()
    assert ref is not None and ref.kind == "reference", "reference build unavailable"
    out = {}
    encoded = {}
    for c in cases.encode_cases():
        s = ref.encode(c["pixels"], c["w"], c["h"], c["ch"], c["cs"])
        assert s is not None, c["name"]
        encoded[c["name"]] = s
        out[f"enc/{c['name']}/stream"] = np.frombuffer(s, dtype=np.uint8)
        out[f"enc/{c['name']}/crc_in"] = np.array([zlib.crc32(c["pixels"].tobytes())], dtype=np.uint32)
    for c in cases.encode_arg_cases():
        dummy = np.zeros(16, dtype=np.uint8)
        s = ref.encode_raw(dummy.ctypes.data, O.QoiDesc(c["w"], c["h"], c["ch"], c["cs"]))
        assert s[0] == 0, c["name"]
        out[f"encarg/{c['name']}/null"] = np.array([1], dtype=np.uint8)
    for c in cases.decode_cases(encoded):
        px, d = ref.decode(c["stream"], c["channels"], c["size"])
        out[f"dec/{c['name']}/stream"] = np.frombuffer(c["stream"], dtype=np.uint8)
        out[f"dec/{c['name']}/ok"] = np.array([px is not None], dtype=np.uint8)
        # desc is written before validation whenever the size/arg checks pass (qoi.h:507-511)
        out[f"dec/{c['name']}/desc"] = np.array([d.width, d.height, d.channels, d.colorspace], dtype=np.uint32)
        out[f"dec/{c['name']}/pixels"] = px if px is not None else np.zeros(0, dtype=np.uint8)
    path = os.path.join(ROOT, "tests", "golden", "qoi_golden.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {len(out)} arrays, {os.path.getsize(path)} bytes")


if __name__ == "__main__":
    main()
