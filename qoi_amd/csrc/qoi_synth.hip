// qoi_synth.hip — counter-based synthetic RGBA frame generator (benchmark/test utility).
// Bit-for-bit the function stated in qoi_amd/synth.py: pixel = f(kind, seed, frame, i, width).
// Lets bench.py build multi-GB frame batches directly in HBM (no PCIe traffic) while the
// CPU baseline re-makes the very same frames on the host.
#include "qoi_dev.h"
#include "qoi_kernels.h"

namespace qoimi {

__device__ __forceinline__ uint32_t mix32(uint32_t x) {      // lowbias32
    x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ uint32_t rnd(uint32_t key, uint32_t idx) { return mix32(idx * 0x9E3779B1u + key); }

__device__ __forceinline__ uint32_t synth_px(int kind, uint32_t key, uint32_t i, uint32_t width) {
    switch (kind) {
        case 0: return rnd(key, i);                                   // noise
        case 3: return rnd(key, 0u);                                  // constant
        case 1: {                                                     // photo
            const uint32_t sel = rnd(key ^ 0xA5A5A5A5u, i);
            const uint32_t j = i - ((sel & 15u) == 0u ? 1u : 0u);
            const uint32_t w = rnd(key, j);
            const uint32_t xr = j % width, yr = j / width;
            const uint32_t base_r = (xr >> 3) + (yr >> 4);
            const uint32_t base_g = (xr >> 4) + (yr >> 3);
            const uint32_t base_b = (xr + yr) >> 4;
            const uint32_t kick = (((w >> 12) & 31u) == 0u) ? ((w >> 16) & 7u) : 0u;
            const uint32_t r = (base_r + (w & 3u)) & 255u;
            const uint32_t g = (base_g + ((w >> 4) & 1u) + kick) & 255u;
            const uint32_t b = (base_b + ((w >> 8) & 1u)) & 255u;
            return r | (g << 8) | (b << 16) | 0xFF000000u;
        }
        default: {                                                    // uiflat
            const uint32_t x = i % width, y = i / width;
            const uint32_t t = rnd(key, (y / 64u) * 4099u + (x / 96u));
            const uint32_t pal = t & 15u;
            const uint32_t col = rnd(key ^ 0x5EED5EEDu, pal);
            const uint32_t alpha = (pal & 1u) ? 128u : 255u;
            return (col & 0x00FFFFFFu) | (alpha << 24);
        }
    }
}

__global__ __launch_bounds__(256) void synth_frames(SynthParams p) {
    const uint32_t frame = blockIdx.y;
    const uint32_t key = mix32(p.seed + (p.first_frame + frame) * 0x85EBCA6Bu);
    uint32_t* __restrict__ dst = reinterpret_cast<uint32_t*>(p.pixels + (size_t)frame * p.pixel_stride);
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < p.npx; i += gridDim.x * 256u)
        dst[i] = synth_px(p.kind, key, i, p.width);
}

void launch_synth(const SynthParams& p, hipStream_t st) {
    uint32_t bx = (p.npx + 255u) / 256u;
    if (bx > 4096u) bx = 4096u;
    hipLaunchKernelGGL(synth_frames, dim3(bx, p.n_frames), dim3(256), 0, st, p);
}

}  // namespace qoimi
