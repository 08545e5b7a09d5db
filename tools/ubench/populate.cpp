// What does it cost to get 33 MB of fresh malloc memory populated (qoi_decode's result)?  g++ -O2 -pthread populate.cpp -o populate
// malloc vs aligned_alloc(2 MiB), MADV_HUGEPAGE or not, 1..4 threads; prints ms per variant (best of 5).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <sys/mman.h>
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void pop(char* p, size_t n, bool huge, int mode) {
    uintptr_t a = ((uintptr_t)p + 4095) & ~(uintptr_t)4095, e = ((uintptr_t)p + n) & ~(uintptr_t)4095;
    if (huge) madvise((void*)a, e - a, MADV_HUGEPAGE);
    if (mode == 0) { if (madvise((void*)a, e - a, MADV_POPULATE_WRITE) != 0) mode = 1; }
    if (mode == 1) for (uintptr_t q = a; q < e; q += 4096) *(volatile char*)q = 0;
}
int main() {
    const size_t n = 3840u * 2160u * 4u;
    FILE* f = fopen("/sys/kernel/mm/transparent_hugepage/enabled", "r"); char buf[128] = "?"; if (f) { fgets(buf, 128, f); fclose(f); }
    printf("THP enabled: %s", buf);
    for (int aligned = 0; aligned < 2; ++aligned) for (int huge = 0; huge < 2; ++huge) for (int mode = 0; mode < 2; ++mode) for (int nt = 1; nt <= 4; ++nt) {
        double best = 1e9;
        for (int r = 0; r < 5; ++r) {
            char* p = aligned ? (char*)aligned_alloc((size_t)2 << 20, (n + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1)) : (char*)malloc(n);
            const double t0 = now();
            std::vector<std::thread> th;
            const size_t part = ((n / nt) + ((size_t)2 << 20) - 1) & ~(((size_t)2 << 20) - 1);
            for (int i = 0; i < nt; ++i) { size_t lo = i * part; if (lo >= n) break; size_t m = lo + part > n ? n - lo : part; th.emplace_back(pop, p + lo, m, huge != 0, mode); }
            for (auto& t : th) t.join();
            const double t1 = now();
            if (t1 - t0 < best) best = t1 - t0;
            free(p);
        }
        printf("aligned=%d hugepage=%d %s threads=%d : %.3f ms\n", aligned, huge, mode ? "touch" : "POPULATE_WRITE", nt, best);
    }
    return 0;
}
