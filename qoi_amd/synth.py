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
  ``photo_hard``  Kodak-like: steeper ramps + 4-bit luma noise shared by r,g,b, 3-bit noise of their own on r and b,
                a red kick on 1 pixel in 16, low-noise 32-pixel patches (1 in 8), 1 repeat in 64
                                                  -> ~2.1 B/px; LUMA 73 % / RGB 12 % / DIFF 9 % / INDEX 4 % / RUN 3 %
  ``sprite_alpha``  256 x 256 sprites of ``photo`` content: opaque disc, an 18-pixel soft edge whose alpha falls
                255 -> 0, fully transparent {0,0,0,0} outside
                                                  -> ~1.6 B/px; 29 % of the chunks QOI_OP_RGBA, the rest the photo mix + runs
"""
from __future__ import annotations

import numpy as np

KINDS = ("noise", "photo", "uiflat", "constant", "photo_hard", "sprite_alpha")
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
        if kind == "sprite_alpha":
            base = frame_u32("photo", width, height, frame, seed)
            dx = (x & np.uint32(255)).astype(np.int64) - 128
            dy = (y & np.uint32(255)).astype(np.int64) - 128
            d2 = (dx * dx + dy * dy).astype(np.uint32)
            a = np.minimum((np.uint32(14000) - np.minimum(d2, np.uint32(14000))) >> np.uint32(4), np.uint32(255))
            return np.where(a == 0, np.uint32(0), (base & np.uint32(0x00FFFFFF)) | (a << np.uint32(24))).astype(np.uint32)
        if kind == "photo_hard":
            w = _rnd(key, i)
            w2 = _rnd(key ^ np.uint32(0x3C3C3C3C), i)
            base_r = (x >> np.uint32(2)) + (y >> np.uint32(3))
            base_g = (x >> np.uint32(3)) + (y >> np.uint32(2))
            base_b = (x + y) >> np.uint32(3)
            lum = w & np.uint32(15)
            nr = (w >> np.uint32(4)) & np.uint32(7)
            nb = (w >> np.uint32(8)) & np.uint32(7)
            kick = (((w >> np.uint32(12)) & np.uint32(15)) == 0).astype(np.uint32)
            kv = (np.uint32(24) + ((w >> np.uint32(16)) & np.uint32(31))) * kick
            blk = _rnd(key ^ np.uint32(0x77777777), (y >> np.uint32(2)) * np.uint32(8191) + (x >> np.uint32(5)))
            smooth = (blk & np.uint32(7)) == 0
            lum = np.where(smooth, lum & np.uint32(1), lum)
            nr = np.where(smooth, nr & np.uint32(1), nr)
            nb = np.where(smooth, nb & np.uint32(1), nb)
            r = (base_r + lum + nr + kv) & np.uint32(255)
            g = (base_g + lum) & np.uint32(255)
            b = (base_b + lum + nb) & np.uint32(255)
            px = r | (g << np.uint32(8)) | (b << np.uint32(16)) | np.uint32(0xFF000000)
            # 1 pixel in 64 repeats its left neighbour's (own) value
            left = np.concatenate([px[:1], px[:-1]])
            return np.where((w2 & np.uint32(63)) == 0, left, px).astype(np.uint32)
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


def stream_hash64(stream) -> int:
    """The 64-bit content hash ``qoimi_hash_streams`` computes on the device (csrc/qoi_synth.hip: hash_streams), restated in
    numpy: sum over the 8-byte little-endian words w_j (the last one zero-padded) of splitmix64(w_j + (j+1) * 0x9E3779B97F4A7C15)."""
    b = np.frombuffer(bytes(stream), dtype=np.uint8)
    pad = (-len(b)) % 8
    if pad:
        b = np.concatenate([b, np.zeros(pad, dtype=np.uint8)])
    w = b.view("<u8").astype(np.uint64)
    with np.errstate(over="ignore"):
        z = w + (np.arange(1, len(w) + 1, dtype=np.uint64) * np.uint64(0x9E3779B97F4A7C15))
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
        return int(z.sum(dtype=np.uint64))
