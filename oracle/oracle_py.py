"""ctypes bindings for the CPU checkers (TEST INFRASTRUCTURE ONLY).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg — never from the product package ``qoi_amd``.

Two libraries, same ABI:
  * ``port``  oracle/liboracle.so        — our C restatement (oracle/qoi_oracle.c)
  * ``ref``   oracle/_ref/libqoiref.so   — the unmodified reference qoi.h compiled by
                                           oracle/Makefile (symbols renamed ref_qoi_*)
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class QoiDesc(ctypes.Structure):
    """qoi_desc of the reference (qoi.h:236-241)."""
    _fields_ = [("width", ctypes.c_uint), ("height", ctypes.c_uint),
                ("channels", ctypes.c_ubyte), ("colorspace", ctypes.c_ubyte)]


def build(quiet: bool = True) -> None:
    """(Re)build liboracle.so and, when /root/reference exists, _ref/libqoiref.so."""
    subprocess.run(["make", "-C", _HERE], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


class _Lib:
    def __init__(self, path: str, prefix: str, kind: str):
        self.kind = kind
        self.path = path
        self._lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
        self._enc = getattr(self._lib, prefix + "qoi_encode")
        self._dec = getattr(self._lib, prefix + "qoi_decode")
        self._enc.restype = ctypes.c_void_p
        self._enc.argtypes = [ctypes.c_void_p, ctypes.POINTER(QoiDesc), ctypes.POINTER(ctypes.c_int)]
        self._dec.restype = ctypes.c_void_p
        self._dec.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.POINTER(QoiDesc), ctypes.c_int]
        self._free = ctypes.CDLL(None).free
        self._free.argtypes = [ctypes.c_void_p]
        self._free.restype = None

    # -- raw pointer forms (used for timing: malloc/free inside the timed region,
    #    like BENCHMARK_FN in qoibench.c:364-376,446-450,471-481)
    def encode_raw(self, ptr: int, desc: QoiDesc) -> Tuple[int, int]:
        n = ctypes.c_int(0)
        p = self._enc(ptr, ctypes.byref(desc), ctypes.byref(n))
        return p or 0, n.value

    def decode_raw(self, ptr: int, size: int, channels: int) -> Tuple[int, QoiDesc]:
        d = QoiDesc()
        p = self._dec(ptr, size, ctypes.byref(d), channels)
        return p or 0, d

    def free(self, p: int) -> None:
        if p:
            self._free(p)

    # -- numpy conveniences
    def encode(self, pixels: np.ndarray, width: int, height: int, channels: int,
               colorspace: int = 0) -> Optional[bytes]:
        pixels = np.ascontiguousarray(pixels, dtype=np.uint8)
        desc = QoiDesc(width, height, channels, colorspace)
        p, n = self.encode_raw(pixels.ctypes.data, desc)
        if not p:
            return None
        try:
            return ctypes.string_at(p, n)
        finally:
            self.free(p)

    def decode(self, stream: bytes, channels: int = 0, size: Optional[int] = None
               ) -> Tuple[Optional[np.ndarray], QoiDesc]:
        buf = (ctypes.c_ubyte * max(len(stream), 1)).from_buffer_copy(stream.ljust(1, b"\0"))
        n = len(stream) if size is None else size
        p, d = self.decode_raw(ctypes.addressof(buf), n, channels)
        if not p:
            return None, d
        try:
            och = channels if channels else d.channels
            nbytes = d.width * d.height * och
            out = np.frombuffer(ctypes.string_at(p, nbytes), dtype=np.uint8).copy()
            return out, d
        finally:
            self.free(p)


def load_port() -> _Lib:
    path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(path):
        build()
    return _Lib(path, "oracle_", "port")


def load_ref() -> Optional[_Lib]:
    """The unmodified reference, or None when oracle/_ref was never built."""
    path = os.path.join(_HERE, "_ref", "libqoiref.so")
    if not os.path.exists(path):
        if os.path.exists("/root/reference/qoi.h"):
            build()
        else:
            return None
    return _Lib(path, "ref_", "reference")


def chunk_histogram(stream: bytes) -> dict:
    """Walk a stream's chunks (qoi.h:544-575 tag rules) and count them by op."""
    b = np.frombuffer(stream, dtype=np.uint8)
    end = len(b) - 8
    p = 14
    h = {"INDEX": 0, "DIFF": 0, "LUMA": 0, "RUN": 0, "RGB": 0, "RGBA": 0}
    while p < end:
        t = int(b[p])
        if t == 0xFE:
            h["RGB"] += 1; p += 4
        elif t == 0xFF:
            h["RGBA"] += 1; p += 5
        elif t >> 6 == 0:
            h["INDEX"] += 1; p += 1
        elif t >> 6 == 1:
            h["DIFF"] += 1; p += 1
        elif t >> 6 == 2:
            h["LUMA"] += 1; p += 2
        else:
            h["RUN"] += 1; p += 1
    return h
