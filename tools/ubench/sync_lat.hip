// How long does hipStreamSynchronize take to return after the last operation of a stream - a kernel, or a small copy back
// to pinned memory behind the kernel, or a kernel that leaves a word in pinned host memory itself?  MI355X: the same (594 / 596 /
// 594 us per nine 66-us kernels + sync): the copy back that ends a decode round is not what its synchronisation waits for.
// build: hipcc -O2 --offload-arch=gfx950 tools/ubench/sync_lat.hip -o /tmp/sync_lat
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void spin(uint32_t* p, int n) { uint32_t v = 0; for (int i = 0; i < n; ++i) v += __builtin_amdgcn_s_memtime() & 1; if (threadIdx.x == 0 && blockIdx.x == 0) p[0] = v; }
__global__ void spin_host(uint32_t* p, uint32_t* h, int n) { uint32_t v = 0; for (int i = 0; i < n; ++i) v += __builtin_amdgcn_s_memtime() & 1; if (threadIdx.x == 0 && blockIdx.x == 0) { p[0] = v; __hip_atomic_store(h, v + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); __threadfence_system(); } }
int main() {
    uint32_t *d, *h; hipMalloc(&d, 256); hipHostMalloc(&h, 256);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    for (int mode = 0; mode < 3; ++mode) for (int rep = 0; rep < 2; ++rep) {
        const int N = 2000;
        for (int w = 0; w < 50; ++w) { hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, d, 2000); hipStreamSynchronize(st); }
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < N; ++i) {
            for (int k = 0; k < 8; ++k) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, d, 2000);
            if (mode == 2) hipLaunchKernelGGL(spin_host, dim3(64), dim3(256), 0, st, d, h, 2000);
            else hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, st, d, 2000);
            if (mode == 1) hipMemcpyAsync(h, d, 12, hipMemcpyDeviceToHost, st);
            hipStreamSynchronize(st);
        }
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("%s: %.1f us per (9 kernels + sync)\n", mode == 0 ? "kernel last            " : mode == 1 ? "kernel + 12-byte copy  " : "kernel writes host word", us);
    }
    return 0;
}
