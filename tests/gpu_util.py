"""Helpers for the -m gpu tests: device buffers via torch (plumbing only)."""
import ctypes

import numpy as np


def torch_mod():
    import torch
    return torch


class DeviceBatch:
    """Owns device buffers for an encode/decode batch of equally shaped images."""

    def __init__(self, ctx, width, height, channels, n):
        torch = torch_mod()
        from qoi_amd import api
        self.ctx, self.w, self.h, self.ch, self.n = ctx, width, height, channels, n
        self.npx = width * height
        self.desc = api.QoiDesc(width, height, channels, 0)
        self.pixel_stride = (self.npx * channels + 255) // 256 * 256
        self.stream_stride = (api.encode_bound(width, height, channels) + 255) // 256 * 256
        self.pixels = torch.zeros(n * self.pixel_stride, dtype=torch.uint8, device="cuda")
        self.streams = torch.zeros(n * self.stream_stride, dtype=torch.uint8, device="cuda")
        self.lens = torch.zeros(n, dtype=torch.int32, device="cuda")
        self.stream = torch.cuda.current_stream().cuda_stream

    def upload(self, i, arr):
        torch = torch_mod()
        a = torch.from_numpy(np.ascontiguousarray(arr).reshape(-1))
        self.pixels[i * self.pixel_stride:i * self.pixel_stride + a.numel()].copy_(a)

    def encode(self):
        self.ctx.encode_batch(self.pixels.data_ptr(), self.pixel_stride, self.desc, self.n,
                              self.streams.data_ptr(), self.stream_stride, self.lens.data_ptr(), self.stream)
        self.ctx.encode_status(self.stream)
        return self.lens.cpu().numpy()

    def stream_bytes(self, i, length):
        return self.streams[i * self.stream_stride:i * self.stream_stride + int(length)].cpu().numpy().tobytes()

    def decode_into(self, out, lens, out_channels=0):
        """Decode this batch's streams into torch uint8 buffer `out` (pixel_stride spacing)."""
        och = out_channels or self.ch
        stride = (self.npx * och + 255) // 256 * 256
        self.ctx.decode_batch(self.streams.data_ptr(), self.stream_stride, [int(x) for x in lens],
                              [self.desc] * self.n, out_channels, out.data_ptr(), stride, self.stream)
        return stride
