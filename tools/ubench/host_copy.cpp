// Micro-benchmark (diagnostic): how fast does one 4K frame (33 MB) get from PAGEABLE host memory to the GPU and a 10 MB stream back?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    const size_t N = 3840 * 2160 * 4, M = 10 << 20;
    char* h = (char*)malloc(N); memset(h, 1, N);
    char* hout = (char*)malloc(M); memset(hout, 1, M);
    char *d, *pin, *pin2; hipMalloc(&d, N); hipHostMalloc(&pin, N); hipHostMalloc(&pin2, N); memset(pin, 2, N); memset(pin2, 2, N);
    hipStream_t st; hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
    auto rep = [&](const char* name, auto fn) { fn(); double best = 1e9; for (int i = 0; i < 5; ++i) { double t0 = now(); fn(); double t = now() - t0; if (t < best) best = t; } printf("%-44s %7.3f ms\n", name, best * 1e3); };
    rep("H2D 33MB hipMemcpy pageable", [&] { hipMemcpy(d, h, N, hipMemcpyHostToDevice); });
    rep("H2D 33MB hipMemcpy pinned", [&] { hipMemcpy(d, pin, N, hipMemcpyHostToDevice); });
    rep("H2D 33MB hipMemcpyAsync pinned + sync", [&] { hipMemcpyAsync(d, pin, N, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); });
    rep("memcpy 33MB pageable -> pinned (1 thread)", [&] { memcpy(pin, h, N); });
    rep("register + H2D + unregister", [&] { hipHostRegister(h, N, hipHostRegisterDefault); hipMemcpyAsync(d, h, N, hipMemcpyHostToDevice, st); hipStreamSynchronize(st); hipHostUnregister(h); });
    for (size_t chunk : {1u << 20, 4u << 20, 8u << 20}) {
        char name[64]; snprintf(name, 64, "pipelined memcpy->pinned->H2D chunk %zuMB", chunk >> 20);
        rep(name, [&] { size_t off = 0; int k = 0; hipEvent_t ev[2]; hipEventCreate(&ev[0]); hipEventCreate(&ev[1]);
            while (off < N) { size_t n = N - off < chunk ? N - off : chunk; char* b = (k & 1) ? pin2 : pin; if (k >= 2) hipEventSynchronize(ev[k & 1]);
                memcpy(b, h + off, n); hipMemcpyAsync(d + off, b, n, hipMemcpyHostToDevice, st); hipEventRecord(ev[k & 1], st); off += n; ++k; }
            hipStreamSynchronize(st); hipEventDestroy(ev[0]); hipEventDestroy(ev[1]); });
    }
    for (int nt : {2, 4}) {
        char name[64]; snprintf(name, 64, "memcpy 33MB pageable -> pinned (%d threads)", nt);
        rep(name, [&] { std::vector<std::thread> th; for (int t = 0; t < nt; ++t) th.emplace_back([&, t] { size_t a = N / nt * t, b = t + 1 == nt ? N : N / nt * (t + 1); memcpy(pin + a, h + a, b - a); }); for (auto& x : th) x.join(); });
    }
    rep("D2H 10MB hipMemcpy -> pageable", [&] { hipMemcpy(hout, d, M, hipMemcpyDeviceToHost); });
    rep("D2H 10MB hipMemcpy -> pinned", [&] { hipMemcpy(pin, d, M, hipMemcpyDeviceToHost); });
    rep("D2H 10MB -> pinned + memcpy to pageable", [&] { hipMemcpy(pin, d, M, hipMemcpyDeviceToHost); memcpy(hout, pin, M); });
    rep("malloc+touch 41MB (reference's worst-case buffer)", [&] { char* p = (char*)malloc(41 << 20); for (size_t i = 0; i < (41u << 20); i += 4096) p[i] = 1; free(p); });
    rep("malloc 41MB untouched + free", [&] { char* p = (char*)malloc(41 << 20); p[0] = 1; free(p); });
    rep("D2H 33MB hipMemcpy -> pageable", [&] { hipMemcpy(h, d, N, hipMemcpyDeviceToHost); });
    rep("D2H 33MB -> fresh malloc (page faults incl.)", [&] { char* p = (char*)malloc(N); hipMemcpy(p, d, N, hipMemcpyDeviceToHost); free(p); });
    return 0;
}
