"""Python host side of the MI355X QOI path — a ctypes binding of libqoi_mi355x.so.

Mirrors the reference's interface for this path (phoboslab/qoi ``qoi.h``):

* :func:`qoi_encode` / :func:`qoi_decode` — same argument meaning and failure
  behaviour (``None`` where the C functions return ``NULL``) as ``qoi.h:278`` /
  ``qoi.h:289``; :class:`QoiDesc` is ``qoi_desc`` (``qoi.h:236-241``).
* :class:`Context` — the additive device-resident batch API (``qoimi_*``), taking
  raw device pointers (e.g. ``torch.Tensor.data_ptr()``) and a HIP stream handle.

There is no CPU fallback here: if the shared library or the GPU is missing the
calls raise / return ``None`` exactly as the C-ABI does.  torch is NOT imported by
this module; bench.py and the tests use it only for device memory and streams.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence, Tuple

import numpy as np

QOI_SRGB = 0
QOI_LINEAR = 1

_HERE = os.path.dirname(os.path.abspath(__file__))
# (tests and measurement tools that want another build of the library - the test flavour, an experimental build - set this before
# the first call: tests/libsel.py; nothing in the environment changes it)
LIB_PATH = os.path.join(_HERE, "lib", "libqoi_mi355x.so")

EXPORTS = (
    # Part 1 — drop-in symbols (qoi.h:252,265,278,289)
    "qoi_encode", "qoi_decode", "qoi_write", "qoi_read",
    # Part 2 — additive
    "qoimi_ctx_create", "qoimi_ctx_destroy", "qoimi_last_error", "qoimi_encode_bound",
    "qoimi_encode_batch", "qoimi_encode_status", "qoimi_decode_batch", "qoimi_synth_frames",
    "qoimi_decode_stats", "qoimi_version", "qoimi_set_profiling", "qoimi_get_profile", "qoimi_kernel_name",
    "qoimi_encode_suspect_calls", "qoimi_encode_retries", "qoimi_set_encode_small_call_order", "qoimi_workspace_bytes", "qoimi_set_decode_record_cap", "qoimi_hash_streams", "qoimi_encode_images",
)


class QoiDesc(ctypes.Structure):
    """``qoi_desc`` (qoi.h:236-241)."""
    _fields_ = [("width", ctypes.c_uint), ("height", ctypes.c_uint),
                ("channels", ctypes.c_ubyte), ("colorspace", ctypes.c_ubyte)]

    def __repr__(self):
        return f"QoiDesc({self.width}x{self.height}, channels={self.channels}, colorspace={self.colorspace})"


class QoiError(RuntimeError):
    pass


_lib = None


def load_library() -> ctypes.CDLL:
    """Load libqoi_mi355x.so (built in-tree by ``__graft_entry__.build()``); raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QoiError(f"{LIB_PATH} not built - run `python -c 'import __graft_entry__ as g; g.build()'` "
                       "(there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    vp, sz, ci = ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int
    lib.qoi_encode.restype = vp
    lib.qoi_encode.argtypes = [vp, ctypes.POINTER(QoiDesc), ctypes.POINTER(ci)]
    lib.qoi_decode.restype = vp
    lib.qoi_decode.argtypes = [vp, ci, ctypes.POINTER(QoiDesc), ci]
    lib.qoi_write.restype = ci
    lib.qoi_write.argtypes = [ctypes.c_char_p, vp, ctypes.POINTER(QoiDesc)]
    lib.qoi_read.restype = vp
    lib.qoi_read.argtypes = [ctypes.c_char_p, ctypes.POINTER(QoiDesc), ci]
    lib.qoimi_ctx_create.restype = ci
    lib.qoimi_ctx_create.argtypes = [ci, ctypes.POINTER(vp)]
    lib.qoimi_ctx_destroy.restype = None
    lib.qoimi_ctx_destroy.argtypes = [vp]
    lib.qoimi_last_error.restype = ctypes.c_char_p
    lib.qoimi_last_error.argtypes = []
    lib.qoimi_version.restype = ctypes.c_char_p
    lib.qoimi_version.argtypes = []
    lib.qoimi_encode_bound.restype = sz
    lib.qoimi_encode_bound.argtypes = [ctypes.POINTER(QoiDesc)]
    lib.qoimi_encode_batch.restype = ci
    lib.qoimi_encode_batch.argtypes = [vp, vp, sz, ctypes.POINTER(QoiDesc), ci, vp, sz, vp, vp]
    lib.qoimi_encode_status.restype = ci
    lib.qoimi_encode_status.argtypes = [vp, vp]
    lib.qoimi_decode_batch.restype = ci
    lib.qoimi_decode_batch.argtypes = [vp, vp, sz, ctypes.POINTER(ci), ctypes.POINTER(QoiDesc), ci, ci, vp, sz, vp]
    lib.qoimi_synth_frames.restype = ci
    lib.qoimi_synth_frames.argtypes = [vp, ci, ctypes.c_uint, ctypes.c_uint, ci, ctypes.c_uint, ctypes.c_uint, vp, sz, vp]
    lib.qoimi_decode_stats.restype = None
    lib.qoimi_decode_stats.argtypes = [vp, ctypes.POINTER(ctypes.c_longlong)]
    lib.qoimi_set_profiling.restype = ci
    lib.qoimi_set_profiling.argtypes = [vp, ci]
    lib.qoimi_get_profile.restype = ci
    lib.qoimi_get_profile.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_longlong), ci]
    lib.qoimi_kernel_name.restype = ctypes.c_char_p
    lib.qoimi_kernel_name.argtypes = [ci]
    lib.qoimi_set_decode_record_cap.restype = ctypes.c_int
    lib.qoimi_set_decode_record_cap.argtypes = [vp, sz, ctypes.c_int]
    lib.qoimi_workspace_bytes.restype = None
    lib.qoimi_workspace_bytes.argtypes = [vp, ctypes.POINTER(sz)]
    lib.qoimi_encode_suspect_calls.restype = ctypes.c_longlong
    lib.qoimi_encode_suspect_calls.argtypes = [vp]
    lib.qoimi_set_encode_small_call_order.restype = ci
    lib.qoimi_set_encode_small_call_order.argtypes = [vp, ci]
    lib.qoimi_encode_retries.restype = ctypes.c_longlong
    lib.qoimi_encode_retries.argtypes = [vp]
    lib.qoimi_encode_images.restype = ci
    lib.qoimi_encode_images.argtypes = [vp, vp, ctypes.POINTER(sz), ctypes.POINTER(QoiDesc), ci, vp, ctypes.POINTER(sz), vp, vp]
    lib.qoimi_hash_streams.restype = ci
    lib.qoimi_hash_streams.argtypes = [vp, vp, sz, vp, ci, vp, vp]
    _lib = lib
    return lib


def _libc_free(p: int) -> None:
    libc = ctypes.CDLL(None)
    libc.free.argtypes = [ctypes.c_void_p]
    libc.free.restype = None
    libc.free(p)


def last_error() -> str:
    return load_library().qoimi_last_error().decode()


def encode_bound(width: int, height: int, channels: int) -> int:
    return int(load_library().qoimi_encode_bound(ctypes.byref(QoiDesc(width, height, channels, 0))))


# ----------------------------------------------------------------------------------
# drop-in functions on host memory
# ----------------------------------------------------------------------------------
def qoi_encode(data, desc: QoiDesc) -> Optional[bytes]:
    """``qoi_encode`` (qoi.h:278): raw RGB/RGBA bytes + desc -> QOI stream, ``None`` on failure."""
    lib = load_library()
    arr = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray))
                               else np.asarray(data, dtype=np.uint8))
    # the C function cannot know how long `data` is (qoi.h:278 trusts width*height*channels); this wrapper can: a short
    # buffer would make the copy engine read past the allocation
    if _desc_plausible(desc) and arr.size < desc.width * desc.height * desc.channels:
        return None
    n = ctypes.c_int(0)
    p = lib.qoi_encode(arr.ctypes.data, ctypes.byref(desc), ctypes.byref(n))
    if not p:
        return None
    try:
        return ctypes.string_at(p, n.value)
    finally:
        _libc_free(p)


def _desc_plausible(desc: QoiDesc) -> bool:
    """qoi.h:364-372's rules - only where they hold does width*height*channels say how many bytes the codec will read."""
    return (desc.width > 0 and desc.height > 0 and desc.channels in (3, 4) and desc.colorspace <= 1
            and desc.height < 400000000 // desc.width)


def qoi_decode(data: bytes, channels: int = 0, size: Optional[int] = None
               ) -> Tuple[Optional[np.ndarray], QoiDesc]:
    """``qoi_decode`` (qoi.h:289): QOI stream -> (pixels uint8[w*h*channels] or ``None``, desc)."""
    lib = load_library()
    buf = (ctypes.c_ubyte * max(len(data), 1)).from_buffer_copy(bytes(data).ljust(1, b"\0"))
    desc = QoiDesc()
    p = lib.qoi_decode(ctypes.addressof(buf), len(data) if size is None else size, ctypes.byref(desc), channels)
    if not p:
        return None, desc
    try:
        och = channels if channels else desc.channels
        return np.frombuffer(ctypes.string_at(p, desc.width * desc.height * och), dtype=np.uint8).copy(), desc
    finally:
        _libc_free(p)


def qoi_write(filename: str, data, desc: QoiDesc) -> int:
    """``qoi_write`` (qoi.h:252): returns bytes written, 0 on failure."""
    arr = np.ascontiguousarray(np.asarray(data, dtype=np.uint8))
    if _desc_plausible(desc) and arr.size < desc.width * desc.height * desc.channels:
        return 0                                     # short pixel buffer: as a failed qoi_write (qoi.h:259-263)
    return int(load_library().qoi_write(filename.encode(), arr.ctypes.data, ctypes.byref(desc)))


def qoi_read(filename: str, channels: int = 0) -> Tuple[Optional[np.ndarray], QoiDesc]:
    """``qoi_read`` (qoi.h:265)."""
    lib = load_library()
    desc = QoiDesc()
    p = lib.qoi_read(filename.encode(), ctypes.byref(desc), channels)
    if not p:
        return None, desc
    try:
        och = channels if channels else desc.channels
        return np.frombuffer(ctypes.string_at(p, desc.width * desc.height * och), dtype=np.uint8).copy(), desc
    finally:
        _libc_free(p)


# ----------------------------------------------------------------------------------
# device-resident batch API
# ----------------------------------------------------------------------------------
class Context:
    """One GPU + workspace (``qoimi_ctx``).  Pointers are raw device addresses (ints)."""

    def __init__(self, device: int = 0):
        self._lib = load_library()
        h = ctypes.c_void_p()
        rc = self._lib.qoimi_ctx_create(device, ctypes.byref(h))
        if rc != 0:
            raise QoiError(f"qoimi_ctx_create({device}) failed ({rc}): {last_error()}")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._lib.qoimi_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int, what: str):
        if rc != 0:
            raise QoiError(f"{what} failed ({rc}): {last_error()}")

    def encode_batch(self, d_pixels: int, pixel_stride: int, desc: QoiDesc, n_images: int,
                     d_streams: int, stream_stride: int, d_stream_len: int, stream: int = 0) -> None:
        self._check(self._lib.qoimi_encode_batch(self._h, d_pixels, pixel_stride, ctypes.byref(desc), n_images,
                                                 d_streams, stream_stride, d_stream_len, stream), "qoimi_encode_batch")

    def encode_images(self, d_pixels: int, pixel_offsets: Sequence[int], descs: Sequence[QoiDesc], d_streams: int,
                      stream_offsets: Sequence[int], d_stream_len: int, stream: int = 0) -> None:
        """Images of different shapes / channel counts in one call (``qoimi_encode_images``)."""
        n = len(descs)
        if len(pixel_offsets) != n or len(stream_offsets) != n:
            raise QoiError("encode_images: one pixel offset and one stream offset per descriptor")
        po = (ctypes.c_size_t * n)(*[int(x) for x in pixel_offsets])
        so = (ctypes.c_size_t * n)(*[int(x) for x in stream_offsets])
        self._check(self._lib.qoimi_encode_images(self._h, d_pixels, po, (QoiDesc * n)(*descs), n, d_streams, so, d_stream_len, stream), "qoimi_encode_images")

    def encode_status(self, stream: int = 0) -> None:
        self._check(self._lib.qoimi_encode_status(self._h, stream), "qoimi_encode_status")

    def encode_suspect_calls(self) -> int:
        """Encode calls of this context made with the exchange probe since the last passed LDS-order check when a repeat failed (0: never)."""
        return int(self._lib.qoimi_encode_suspect_calls(self._h))

    def set_encode_small_call_order(self, by_workgroup_index: bool) -> None:
        """Calls of fewer than 8 images: units by ticket (default, safe on a shared device) or by workgroup index (4 us less per 4K
        frame; then ``encode_status`` must be called before the streams are read)."""
        self._check(self._lib.qoimi_set_encode_small_call_order(self._h, 1 if by_workgroup_index else 0), "qoimi_set_encode_small_call_order")

    def encode_retries(self) -> int:
        """Calls ``encode_status`` encoded again order-free because a placement wait had given up (0: never)."""
        return int(self._lib.qoimi_encode_retries(self._h))

    def decode_batch(self, d_streams: int, stream_stride: int, sizes: Sequence[int], descs: Sequence[QoiDesc],
                     channels: int, d_pixels: int, pixel_stride: int, stream: int = 0) -> None:
        n = len(sizes)
        if len(descs) != n:
            raise QoiError(f"decode_batch: {n} sizes but {len(descs)} descriptors")
        if isinstance(sizes, ctypes.Array) and isinstance(descs, ctypes.Array):
            c_sizes, c_descs = sizes, descs               # the caller's own C arrays (int[n], qoi_desc[n]): handed through as they are
        else:
            if n > 1 and any(int(sz) > stream_stride for sz in sizes):
                raise QoiError("decode_batch: a stream is longer than stream_stride")
            c_sizes = (ctypes.c_int * n)(*[int(s) for s in sizes])
            c_descs = (QoiDesc * n)(*descs)
        self._check(self._lib.qoimi_decode_batch(self._h, d_streams, stream_stride, c_sizes, c_descs, n, channels,
                                                 d_pixels, pixel_stride, stream), "qoimi_decode_batch")

    def synth_frames(self, kind: int, seed: int, first_frame: int, n_frames: int, width: int, height: int,
                     d_pixels: int, pixel_stride: int, stream: int = 0) -> None:
        self._check(self._lib.qoimi_synth_frames(self._h, kind, seed, first_frame, n_frames, width, height,
                                                 d_pixels, pixel_stride, stream), "qoimi_synth_frames")

    def hash_streams(self, d_streams: int, stream_stride: int, d_stream_len: int, n_streams: int, d_hash: int, stream: int = 0) -> None:
        """64-bit content hash of every stream into the device array d_hash (uint64[n_streams]); synth.stream_hash64 is the same function."""
        self._check(self._lib.qoimi_hash_streams(self._h, d_streams, stream_stride, d_stream_len, n_streams, d_hash, stream), "qoimi_hash_streams")

    def set_profiling(self, on: bool) -> None:
        self._check(self._lib.qoimi_set_profiling(self._h, 1 if on else 0), "qoimi_set_profiling")

    def get_profile(self, stream: int = 0) -> dict:
        """kernel name -> (accumulated ms, launches) since profiling was enabled (syncs stream)."""
        ms = (ctypes.c_double * 64)()
        calls = (ctypes.c_longlong * 64)()
        n = min(64, self._lib.qoimi_get_profile(self._h, stream, ms, calls, 64))
        return {self._lib.qoimi_kernel_name(i).decode(): (ms[i], calls[i]) for i in range(1, n)}

    def workspace_bytes(self) -> dict:
        """Device bytes the context's arenas hold: encode workspace, decode workspace, staging of the host-pointer entry points."""
        out = (ctypes.c_size_t * 3)()
        self._lib.qoimi_workspace_bytes(self._h, out)
        return {"encode": int(out[0]), "decode": int(out[1]), "staging": int(out[2])}

    def set_decode_record_cap(self, nbytes: int, release: bool = False) -> None:
        """Caps the chunk-record arena of decode calls (larger calls run as sub-batches of whole images); release=True frees the arena grown so far."""
        self._check(self._lib.qoimi_set_decode_record_cap(self._h, int(nbytes), 1 if release else 0), "qoimi_set_decode_record_cap")

    def decode_stats(self) -> dict:
        out = (ctypes.c_longlong * 4)()
        self._lib.qoimi_decode_stats(self._h, out)
        return {"rounds": out[0], "redo_segments": out[1], "segments": out[2], "sync_fallback_segments": out[3]}
