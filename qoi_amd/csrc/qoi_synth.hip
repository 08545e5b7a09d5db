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

__device__ __forceinline__ uint32_t photo_px(uint32_t key, uint32_t i, uint32_t width) {
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
// photo_hard before the 1-in-64 repeat (synth.py)
__device__ __forceinline__ uint32_t hard_px(uint32_t key, uint32_t i, uint32_t width) {
    const uint32_t x = i % width, y = i / width;
    const uint32_t w = rnd(key, i);
    const uint32_t blk = rnd(key ^ 0x77777777u, (y >> 2) * 8191u + (x >> 5));
    const bool smooth = (blk & 7u) == 0u;
    uint32_t lum = w & 15u, nr = (w >> 4) & 7u, nb = (w >> 8) & 7u;
    if (smooth) { lum &= 1u; nr &= 1u; nb &= 1u; }
    const uint32_t kv = (((w >> 12) & 15u) == 0u) ? 24u + ((w >> 16) & 31u) : 0u;
    const uint32_t r = ((x >> 2) + (y >> 3) + lum + nr + kv) & 255u;
    const uint32_t g = ((x >> 3) + (y >> 2) + lum) & 255u;
    const uint32_t b = (((x + y) >> 3) + lum + nb) & 255u;
    return r | (g << 8) | (b << 16) | 0xFF000000u;
}

__device__ __forceinline__ uint32_t synth_px(int kind, uint32_t key, uint32_t i, uint32_t width) {
    switch (kind) {
        case 4: {                                                     // photo_hard
            const uint32_t w2 = rnd(key ^ 0x3C3C3C3Cu, i);
            return hard_px(key, ((w2 & 63u) == 0u && i != 0u) ? i - 1u : i, width);
        }
        case 5: {                                                     // sprite_alpha: photo content under a soft-edged disc per 256 x 256 tile
            const uint32_t x = i % width, y = i / width;
            const int dx = (int)(x & 255u) - 128, dy = (int)(y & 255u) - 128;
            const uint32_t d2 = (uint32_t)(dx * dx + dy * dy);
            const uint32_t a = min((14000u - min(d2, 14000u)) >> 4, 255u);
            if (a == 0u) return 0u;
            return (photo_px(key, i, width) & 0x00FFFFFFu) | (a << 24);
        }
        case 0: return rnd(key, i);                                   // noise
        case 3: return rnd(key, 0u);                                  // constant
        case 1: return photo_px(key, i, width);                       // photo
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

// 64-bit content hash of every stream (benchmark / test utility: byte identity of whole batches against hashes of the reference
// encoder's streams without moving the streams to the host).  h = sum over the stream's 8-byte little-endian words w_j (the last one
// zero-padded) of mix64(w_j + (j + 1) * 0x9E3779B97F4A7C15) modulo 2^64, mix64 = the splitmix64 finaliser; tools and tests restate it
// in numpy (qoi_amd/synth.py: stream_hash64).
__device__ __forceinline__ u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31);
}
constexpr uint32_t kHashSlices = 16;
__global__ __launch_bounds__(256) void hash_streams(const uint8_t* __restrict__ streams, size_t stride, const int* __restrict__ lens, u64* __restrict__ out) {
    __shared__ u64 s_part[4];
    const uint32_t img = blockIdx.x / kHashSlices, slice = blockIdx.x % kHashSlices;
    const uint8_t* __restrict__ s = streams + (size_t)img * stride;
    const int len_i = lens[img];
    const u64 len = len_i > 0 ? (u64)len_i : 0ull, nwords = (len + 7u) >> 3;
    u64 acc = 0;
    for (u64 j = (u64)slice * 256u + threadIdx.x; j < nwords; j += (u64)kHashSlices * 256u) {
        u64 w = 0;
        const u64 left = len - j * 8u;
        if (left >= 8u) __builtin_memcpy(&w, s + j * 8u, 8);
        else for (u64 k = 0; k < left; ++k) w |= (u64)s[j * 8u + k] << (8u * k);
        acc += mix64(w + (j + 1u) * 0x9E3779B97F4A7C15ull);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += (u64)__shfl_xor((unsigned long long)acc, o);
    if ((threadIdx.x & 63u) == 0u) s_part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd((unsigned long long*)&out[img], (unsigned long long)(s_part[0] + s_part[1] + s_part[2] + s_part[3]));
}
void launch_hash_streams(const uint8_t* streams, size_t stride, const int* lens, uint32_t n, u64* out, hipStream_t st) {
    (void)hipMemsetAsync(out, 0, (size_t)n * sizeof(u64), st);
    hipLaunchKernelGGL(hash_streams, dim3(n * kHashSlices), dim3(256), 0, st, streams, stride, lens, out);
}

void launch_synth(const SynthParams& p, hipStream_t st) {
    uint32_t bx = (p.npx + 255u) / 256u;
    if (bx > 4096u) bx = 4096u;
    hipLaunchKernelGGL(synth_frames, dim3(bx, p.n_frames), dim3(256), 0, st, p);
}

}  // namespace qoimi
