#!/usr/bin/env python3
"""qoiconv on the MI355X library: convert between png <> qoi.

Mirrors the reference's ``qoiconv.c``: same command line (``qoiconv.c:34-41``), same rules - a PNG whose own
format is not 3-channel is loaded as RGBA (``qoiconv.c:51-54``), ``.qoi`` files go through ``qoi_read`` /
``qoi_write`` (``qoiconv.c:58-63, 78-84``) - here the ones of ``libqoi_mi355x.so``, i.e. the GPU path behind the
reference's C-ABI - colourspace is written as QOI_SRGB (``qoiconv.c:83``), exit status 1 on any failure.
PNG I/O is ``tools/png_io.py`` (the reference uses stb_image / stb_image_write, third-party headers that are not
part of its repository).

  python tools/qoiconv_mi355x.py input.png output.qoi
  python tools/qoiconv_mi355x.py input.qoi output.png
"""
from __future__ import annotations

import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import png_io  # noqa: E402


def main(argv) -> int:
    if len(argv) < 3:
        print("Usage: qoiconv <infile> <outfile>")
        print("Examples:")
        print("  qoiconv input.png output.qoi")
        print("  qoiconv input.qoi output.png")
        return 1
    src, dst = argv[1], argv[2]
    from qoi_amd import api          # fails loudly without the library / a gfx950 GPU: there is no CPU codec

    pixels = None
    w = h = channels = 0
    if src.endswith(".png"):
        try:
            data = open(src, "rb").read()
            w, h, channels = png_io.png_info(data)
        except (OSError, png_io.PngError):
            print(f"Couldn't read header {src}")
            return 1
        if channels != 3:                # force all odd encodings to be RGBA (qoiconv.c:51-54)
            channels = 4
        try:
            pixels, w, h = png_io.read_png(data, channels)
        except png_io.PngError:
            pixels = None
    elif src.endswith(".qoi"):
        pixels, desc = api.qoi_read(src, 0)
        if pixels is not None:
            w, h, channels = desc.width, desc.height, desc.channels
            pixels = pixels.reshape(h, w, channels)
    if pixels is None:
        print(f"Couldn't load/decode {src}")
        return 1

    encoded = 0
    if dst.endswith(".png"):
        try:
            blob = png_io.write_png(pixels)
            with open(dst, "wb") as f:
                f.write(blob)
            encoded = len(blob)
        except (OSError, png_io.PngError):
            encoded = 0
    elif dst.endswith(".qoi"):
        encoded = api.qoi_write(dst, pixels, api.QoiDesc(w, h, channels, api.QOI_SRGB))
    if not encoded:
        print(f"Couldn't write/encode {dst}")
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv))
