"""Synthetic RGBA frame generator — host (numpy) statement of the spec.

The reference prescribes no value distribution; these content
classes are ours and are FIXED here.  The device generator in
``csrc/qoi_synth.hip`` (C-ABI ``qoimi_synth_frames``) implements the same
function of ``(seed, frame, pixel_index)`` bit-for-bit, so a frame can be made
on the GPU (no PCIe traffic) and re-made on the host for the CPU baseline and
for parity checks.

Pixel value = f(kind, seed, frame, i, width) with i the row-major pixel index.
All arithmetic is uint32 wrap-around.

kinds
  ``noise``     4 uniform random bytes            -> 5.000 B/px, every chunk QOI_OP_RGBA
  ``photo``     two slow ramps + small correlated noise, alpha 255
                                                  -> ~1.2 B/px, DIFF/LUMA/INDEX/RUN mix
  ``uiflat``    96x64 tiles from a 16-colour palette with two alpha levels
                                                  -> ~0.1 B/px, RUN/INDEX/RGB(A)
  ``constant``  one colour per frame              -> 0.016 B/px, all QOI_OP_RUN
"""
from __future__ import annotations

import numpy as np

KINDS = ("noise", "photo", "uiflat", "constant")
KIND_ID = {k: i for i, k in enumerate(KINDS)}
DEFAULT_SEED = 12345

_M1 = np.uint32(0x7FEB352D)
_M2 = np.uint32(0x846CA68B)
_G = np.uint32(0x9E3779B1)
_F = np.uint32(0x85EBCA6B)


def mix32(x: np.ndarray) -> np.ndarray:
    """lowbias32 integer finaliser (uint32 -> uint32)."""
    x = np.asarray(x, dtype=np.uint32).copy()
    x ^= x >> np.uint32(16)
    x *= _M1
    x ^= x >> np.uint32(15)
    x *= _M2
    x ^= x >> np.uint32(16)
    return x


def _key(seed: int, frame: int) -> np.uint32:
    with np.errstate(over="ignore"):
        return mix32(np.uint32(seed) + np.uint32(frame) * _F)[()]


def _rnd(key: np.uint32, idx: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        return mix32(np.asarray(idx, dtype=np.uint32) * _G + key)


def frame_u32(kind: str, width: int, height: int, frame: int = 0,
              seed: int = DEFAULT_SEED) -> np.ndarray:
    """Return one frame as uint32[height*width], little-endian r,g,b,a bytes."""
    n = width * height
    i = np.arange(n, dtype=np.uint32)
    key = _key(seed, frame)
    with np.errstate(over="ignore"):
        if kind == "noise":
            return _rnd(key, i)
        if kind == "constant":
            c = _rnd(key, np.uint32(0))
            return np.full(n, c, dtype=np.uint32)
        x = i % np.uint32(width)
        y = i // np.uint32(width)
        if kind == "photo":
            # Correlated noise: a pixel reuses its left neighbour's random word (and ramp
            # position) with probability 1/16, which yields true repeats (-> QOI_OP_RUN).
            sel = _rnd(key ^ np.uint32(0xA5A5A5A5), i)
            j = i - ((sel & np.uint32(15)) == 0).astype(np.uint32)
            w = _rnd(key, j)
            xr = j % np.uint32(width)
            yr = j // np.uint32(width)
            base_r = (xr >> np.uint32(3)) + (yr >> np.uint32(4))
            base_g = (xr >> np.uint32(4)) + (yr >> np.uint32(3))
            base_b = ((xr + yr) >> np.uint32(4))
            # 2-bit noise on r, 1-bit on g and b; 1 pixel in 32 gets a green kick of 0..7
            # (-> QOI_OP_LUMA / occasional QOI_OP_RGB)
            kick = ((w >> np.uint32(12)) & np.uint32(31)) == 0
            r = (base_r + (w & np.uint32(3))) & np.uint32(255)
            g = (base_g + ((w >> np.uint32(4)) & np.uint32(1))
                 + kick.astype(np.uint32) * ((w >> np.uint32(16)) & np.uint32(7))) & np.uint32(255)
            b = (base_b + ((w >> np.uint32(8)) & np.uint32(1))) & np.uint32(255)
            return r | (g << np.uint32(8)) | (b << np.uint32(16)) | np.uint32(0xFF000000)
        if kind == "uiflat":
            tx = x // np.uint32(96)
            ty = y // np.uint32(64)
            t = _rnd(key, ty * np.uint32(4099) + tx)
            pal = t & np.uint32(15)
            col = _rnd(key ^ np.uint32(0x5EED5EED), pal)      # 16 palette colours per frame
            alpha = np.where((pal & np.uint32(1)) == 0, np.uint32(255), np.uint32(128))
            return (col & np.uint32(0x00FFFFFF)) | (alpha << np.uint32(24))
    raise ValueError(f"unknown kind {kind!r}")


def frame_rgba(kind: str, width: int, height: int, frame: int = 0,
               seed: int = DEFAULT_SEED) -> np.ndarray:
    """uint8[height, width, 4] view of :func:`frame_u32`."""
    return frame_u32(kind, width, height, frame, seed).view(np.uint8).reshape(height, width, 4)


def frame_rgb(kind: str, width: int, height: int, frame: int = 0,
              seed: int = DEFAULT_SEED) -> np.ndarray:
    """3-channel variant (alpha dropped) for channels==3 tests."""
    return np.ascontiguousarray(frame_rgba(kind, width, height, frame, seed)[..., :3])
