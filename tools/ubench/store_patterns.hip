// Micro-benchmark (diagnostic): how fast can 64-lane wavefronts write per-lane output streams?
//   A  each lane 16 B per store at its own stream position (64 scattered 16-byte pieces per instruction)  - dec_segments today
//   B  4 adjacent lanes write one owner's 64 contiguous bytes (16 owners per instruction, 4 instructions)     - cooperative drain
//   C  each lane 4 consecutive 16-byte stores (64 B burst per lane)
//   D  8 adjacent lanes write one owner's 128 contiguous bytes
// Streams: lane stride kSeg bytes (one decode segment's pixels), wave stride 64*kSeg; every byte written exactly once.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
constexpr uint32_t kSeg = 6912;      // bytes of pixels per lane stream (multiple of 128)
template <int MODE>
__global__ __launch_bounds__(64) void k(uint8_t* out, uint32_t spin) {
    const uint32_t lane = threadIdx.x;
    uint8_t* wbase = out + (size_t)blockIdx.x * 64u * kSeg;
    uint4 v = make_uint4(lane, blockIdx.x, 3, 4);
    for (uint32_t i = 0; i < kSeg / 64u; ++i) {
        // some ALU work between drains, like the decoder's steps (keeps the issue pattern comparable)
        for (uint32_t s = 0; s < spin; ++s) { v.x = v.x * 1664525u + 1013904223u; asm volatile("" : "+v"(v.x)); }
        if (MODE == 0) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                for (uint32_t s = 0; s < spin; ++s) { v.y = v.y * 1664525u + 1013904223u; asm volatile("" : "+v"(v.y)); }
                *reinterpret_cast<uint4*>(wbase + (size_t)lane * kSeg + 64u * i + 16u * j) = v;
            }
        } else if (MODE == 1) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) {
                const uint32_t owner = (lane >> 2) + 16u * j, piece = lane & 3u;
                *reinterpret_cast<uint4*>(wbase + (size_t)owner * kSeg + 64u * i + 16u * piece) = v;
            }
        } else if (MODE == 2) {
#pragma unroll
            for (uint32_t j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(wbase + (size_t)lane * kSeg + 64u * i + 16u * j) = v;
        } else {
            if ((i & 1u) == 0u) {
#pragma unroll
                for (uint32_t j = 0; j < 8; ++j) {
                    const uint32_t owner = (lane >> 3) + 8u * j, piece = lane & 7u;
                    *reinterpret_cast<uint4*>(wbase + (size_t)owner * kSeg + 64u * i + 16u * piece) = v;
                }
            }
        }
    }
}
int main() {
    const uint32_t waves = 256 * 8 * 10;
    const size_t bytes = (size_t)waves * 64 * kSeg;
    uint8_t* d; hipMalloc(&d, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (uint32_t spin : {0u, 40u, 150u}) {
        for (int mode = 0; mode < 4; ++mode) {
            float best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(waves), dim3(64), 0, 0, d, spin);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(waves), dim3(64), 0, 0, d, spin);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(waves), dim3(64), 0, 0, d, spin);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(waves), dim3(64), 0, 0, d, spin);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            printf("spin %3u mode %c: %.3f ms  %.0f GB/s (%.2f GB)\n", spin, "ABCD"[mode], best, bytes / best / 1e6, bytes / 1e9);
        }
    }
    return 0;
}
