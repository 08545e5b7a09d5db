// lds_order_probe.hip — measures (does not assume) how gfx950's LDS serialises the lanes of
// ONE ds_wrxchg_rtn_b32 that hit the same address.  If conflicting lanes are always
// served in ascending lane order, "exchange my pixel into table[slot] and look at what
// came back" is exactly the reference encoder's sequential index probe/update
// (qoi.h:430-436) for 64 pixels at once.  Prints the number of patterns whose result
// differs from the ascending-lane-order model (0 = the fast probe is usable here).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void probe(const uint32_t* slots, const uint64_t* active, uint32_t* out, int patterns) {
    __shared__ uint32_t tab[64];
    const int lane = threadIdx.x & 63;
    for (int p = blockIdx.x; p < patterns; p += gridDim.x) {
        tab[lane] = 0xFFFF0000u | lane;
        __builtin_amdgcn_wave_barrier();
        const uint32_t s = slots[p * 64 + lane];
        const bool on = (active[p] >> lane) & 1ull;
        uint32_t old = 0xDEADBEEFu;
        if (on) old = __hip_atomic_exchange(&tab[s], (uint32_t)(p * 64 + lane), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_wave_barrier();
        out[(size_t)p * 128 + lane] = old;
        out[(size_t)p * 128 + 64 + lane] = tab[lane];
        __builtin_amdgcn_wave_barrier();
    }
}

int main() {
    const int P = 20000;
    std::vector<uint32_t> slots(P * 64); std::vector<uint64_t> act(P);
    srand(1234);
    for (int p = 0; p < P; ++p) {
        const int mode = p % 5;
        const int nb = mode == 0 ? 64 : mode == 1 ? 1 : mode == 2 ? 2 : mode == 3 ? 8 : 32;
        for (int l = 0; l < 64; ++l) slots[p * 64 + l] = (uint32_t)(rand() % nb) * (64 / nb) % 64;
        uint64_t a = 0; for (int l = 0; l < 64; ++l) if (rand() % 8 != 0 || mode == 1) a |= 1ull << l;
        act[p] = a;
    }
    uint32_t *d_s, *d_o; uint64_t* d_a;
    hipMalloc(&d_s, slots.size() * 4); hipMalloc(&d_a, act.size() * 8); hipMalloc(&d_o, (size_t)P * 128 * 4);
    hipMemcpy(d_s, slots.data(), slots.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(d_a, act.data(), act.size() * 8, hipMemcpyHostToDevice);
    std::vector<uint32_t> out((size_t)P * 128);
    long bad = 0, bad_final = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(probe, dim3(rep == 0 ? 1 : 1024), dim3(64), 0, 0, d_s, d_a, d_o, P);
        hipMemcpy(out.data(), d_o, out.size() * 4, hipMemcpyDeviceToHost);
        for (int p = 0; p < P; ++p) {
            uint32_t tab[64]; for (int l = 0; l < 64; ++l) tab[l] = 0xFFFF0000u | l;
            bool ok = true, okf = true;
            for (int l = 0; l < 64; ++l) {
                if (!((act[p] >> l) & 1)) continue;
                const uint32_t s = slots[p * 64 + l];
                if (out[(size_t)p * 128 + l] != tab[s]) ok = false;
                tab[s] = (uint32_t)(p * 64 + l);
            }
            for (int l = 0; l < 64; ++l) if (out[(size_t)p * 128 + 64 + l] != tab[l]) okf = false;
            bad += !ok; bad_final += !okf;
        }
    }
    printf("lds_order_probe: patterns=%d x3 launches, exchange-order mismatches=%ld, final-table mismatches=%ld\n", P, bad, bad_final);
    return 0;
}
