// What a matrix instruction costs a VALU-bound wavefront on gfx950: W wavefronts per SIMD each run iterations of 32 independent
// v_add3_u32 (a VOP3 op: 4 cycles of the SIMD each) with M matrix instructions mixed in, their results read by vector
// instructions 16+ instructions later.  Reported: ticks of s_memtime per iteration per SIMD (= per wavefront / W) and the
// difference to the VALU-only iteration in "VALU slots" (one v_add3 = 1).  Question behind it: does the matrix pipe run beside
// the vector pipe of the SAME SIMD (the encoder's CLS 1 experiment)?
// build: hipcc --offload-arch=gfx950 -O2 -o build/ubench/mfma_mix tools/ubench/mfma_mix.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int kIters = 2000;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
#define V8 asm volatile("v_add3_u32 %0, %0, %8, %9\n v_add3_u32 %1, %1, %8, %9\n v_add3_u32 %2, %2, %8, %9\n v_add3_u32 %3, %3, %8, %9\n" \
                        "v_add3_u32 %4, %4, %8, %9\n v_add3_u32 %5, %5, %8, %9\n v_add3_u32 %6, %6, %8, %9\n v_add3_u32 %7, %7, %8, %9" \
                        : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c));
#define M16(d, a) asm volatile("v_mfma_i32_16x16x32_i8 %0, %1, %2, 2" : "=v"(d) : "v"(a), "v"(pb));
#define M16K64(d, a) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, 2" : "=v"(d) : "v"(a4), "v"(pb4));
#define M4(d, a) asm volatile("v_mfma_i32_4x4x4_16b_i8 %0, %1, %2, 2" : "=v"(d) : "v"(a1), "v"(pb1));
#define M32(d, a) asm volatile("v_mfma_i32_32x32x16_i8 %0, %1, %2, 0" : "=v"(d) : "v"(a), "v"(pb));
#define USE4(d) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(d.x), "v"(d.w));

template <int T>
__global__ __launch_bounds__(256) void k(uint32_t* out, uint64_t* cyc, uint32_t seed) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t r0 = seed + lane, r1 = seed ^ lane, r2 = lane * 3, r3 = 7, r4 = 9, r5 = 11, r6 = 13, r7 = 15;
    const uint32_t b = seed * 3 + 1, c = seed ^ 0x55;
    long a = (long)lane * 0x0101010101010101l, a2 = a + 3, a3 = a + 5, pb = (long)seed * 0x0102030405060708l + lane;
    uint32_t pbl = seed + lane, pbh = seed * 7u;
    asm volatile("" : "+v"(pbl), "+v"(pbh));
    v4i a4 = {(int)lane, 3, 5, 7}, pb4 = {(int)seed, 1, 2, 3};
    int a1 = (int)lane * 0x01010101, pb1 = (int)seed * 0x01020304;
    asm volatile("" : "+v"(a), "+v"(a2), "+v"(a3), "+v"(pb), "+v"(a4), "+v"(pb4), "+v"(a1), "+v"(pb1));
    v4i d1 = {0, 0, 0, 0}, d2 = d1, d3 = d1;
    __shared__ uint32_t lds[4][128];
    const uint32_t la = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&lds[threadIdx.x >> 6][0] + lane;
    uint64_t m1 = 0x5555555555555555ull + seed, m2 = 0x0F0F0F0F0F0F0F0Full + seed, sm = 0;
    asm volatile("" : "+s"(m1), "+s"(m2));
    v16i e = {0};
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
        if (T == 0) { V8 V8 V8 V8 }
        if (T == 1) { M16(d1, a) M16(d2, a2) M16(d3, a3) V8 V8 USE4(d1) USE4(d2) USE4(d3) V8 V8 }          // three back to back, read 16 later
        if (T == 2) { M16(d1, a) V8 M16(d2, a2) V8 M16(d3, a3) V8 USE4(d1) USE4(d2) V8 USE4(d3) }          // spread
        if (T == 3) { M16(d1, a) V8 V8 USE4(d1) V8 V8 }                                                   // one
        if (T == 4) { M32(e, a) V8 V8 V8 asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(e[0]), "v"(e[15])); V8 }   // one 32x32x16: sixteen rows
        if (T == 5) { M4(d1, a) M4(d2, a) M4(d3, a) V8 V8 USE4(d1) USE4(d2) USE4(d3) V8 V8 }              // three 4x4x4
        if (T == 6) { M16K64(d1, a) M16K64(d2, a) M16K64(d3, a) V8 V8 USE4(d1) USE4(d2) USE4(d3) V8 V8 }  // three 16x16x64
        if (T == 7) { M16(d1, a) M16(d2, a2) M16(d3, a3) }                                                // matrix only
        if (T == 8) { M32(e, a) }
        if (T == 9) { M4(d1, a) M4(d2, a) M4(d3, a) }
        if (T == 10) { M16K64(d1, a) M16K64(d2, a) M16K64(d3, a) }
        if (T == 11) { M16(d1, a) M16(d2, a2) V8 V8 USE4(d1) USE4(d2) V8 V8 }                             // two
        if (T == 20) { M32(e, a) V8 asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(e[0]), "v"(e[15])); V8 V8 V8 }    // read after 8
        if (T == 21) { M32(e, a) V8 V8 asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(e[0]), "v"(e[15])); V8 V8 }    // read after 16
        if (T == 22) { M16(d1, a) V8 M16(d2, a2) V8 USE4(d1) M16(d3, a3) V8 USE4(d2) V8 USE4(d3) }        // 8 apart, read 16+ later
        if (T == 23) { asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(pbl) : "s"(0x00FFFFFFu), "v"(c)); pb = (long)(((unsigned long)pbh << 32) | pbl);
                       M32(e, a) V8 V8 V8 asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(e[0]), "v"(e[15])); V8 }   // B operand fresh from a VALU
        if (T == 24) { M32(e, a) V8 V8 V8 asm volatile("v_or3_b32 %0, %1, %2, %3\n v_or3_b32 %0, %4, %5, %0\n v_perm_b32 %0, %6, %7, %0\n v_add3_u32 %0, %0, %8, %9" : "+v"(r0) : "v"(e[0]), "v"(e[1]), "v"(e[2]), "v"(e[4]), "v"(e[5]), "v"(e[8]), "v"(e[9]), "v"(e[10]), "v"(e[15])); V8 }   // several reads
        // what follows the matrix instruction in the encoder's step: exec switches around LDS byte stores, v_mbcnt, compares into SGPRs
#define EXECSW asm volatile("s_mov_b64 exec, %2\n v_add3_u32 %0, %0, %1, %1\n s_mov_b64 exec, %3\n v_add3_u32 %0, %0, %1, %1\n s_mov_b64 exec, -1" : "+v"(r1) : "v"(b), "s"(m1), "s"(m2));
#define LDSW asm volatile("ds_write_b8 %0, %1\n ds_write_b8_d16_hi %0, %1 offset:1" : : "v"(la), "v"(r2) : "memory");
#define MBCNT asm volatile("v_mbcnt_lo_u32_b32 %0, %1, %0\n v_mbcnt_hi_u32_b32 %0, %2, %0" : "+v"(r3) : "s"((uint32_t)m1), "s"((uint32_t)(m1 >> 32)));
#define CMPS asm volatile("v_cmp_lt_u32_e64 %1, %0, %2\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %2, %1" : "+v"(r4), "=&s"(sm) : "v"(b));
        if (T == 30) { M32(e, a) EXECSW V8 V8 V8 asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(e[0]), "v"(e[15])); V8 }
        if (T == 40) { EXECSW V8 V8 V8 V8 }
        if (T == 31) { M32(e, a) LDSW V8 V8 V8 asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(e[0]), "v"(e[15])); V8 }
        if (T == 41) { LDSW V8 V8 V8 V8 }
        if (T == 32) { M32(e, a) CMPS V8 V8 V8 asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(e[0]), "v"(e[15])); V8 }
        if (T == 42) { CMPS V8 V8 V8 V8 }
        if (T == 33) { M32(e, a) MBCNT MBCNT V8 V8 V8 asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(e[0]), "v"(e[15])); V8 }
        if (T == 43) { MBCNT MBCNT V8 V8 V8 V8 }
        if (T == 34) { M32(e, a) asm volatile("s_cmp_lg_u64 %0, 0\n s_cbranch_scc0 1f\n 1:" : : "s"(m1) : "scc"); V8 V8 V8 asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(r0) : "v"(e[0]), "v"(e[15])); V8 }
        if (T == 12) { M4(d1, a) M4(d2, a) M4(d3, a) M4(d1, a) M4(d2, a) M4(d3, a) V8 V8 USE4(d1) USE4(d2) USE4(d3) V8 V8 }   // six 4x4x4
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 256 + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7 + d1.x + d2.y + d3.z + e[3];
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
static double base[8];
template <int T> void run(const char* name, int nvalu) {
    printf("%-44s", name);
    int wi = 0;
    for (int w : {1, 2, 4, 5, 6}) {
        const int blocks = 256 * w, waves = blocks * 4;      // 256-thread workgroups: a wavefront on every SIMD, w workgroups per CU
        uint32_t* out; uint64_t* cyc;
        hipMalloc(&out, waves * 64 * 4); hipMalloc(&cyc, waves * 8);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, out, cyc, 12345u);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(256), 0, 0, out, cyc, 12345u);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(waves); hipMemcpy(h.data(), cyc, waves * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v; s /= waves;
        const double per_simd = s / kIters / w;
        if (T == 0) base[wi] = per_simd;
        // extra cost over the VALU-only iteration scaled to the iteration's own VALU count, in slots of one v_add3 (base / 32)
        const double slot = base[wi] / 32.0;
        printf("  W=%d: %7.1f (%+5.1f slots)", w, per_simd, (per_simd - slot * nvalu) / slot);
        hipFree(out); hipFree(cyc); ++wi;
    }
    printf("\n");
}
int main() {
    printf("ticks per iteration per SIMD (and the cost of the matrix instructions + the reads of their results beyond the iteration's v_add3, in v_add3 slots)\n");
    run<0>("32 v_add3", 32);
    run<3>("1 x 16x16x32 i8 + 32 v_add3 (+1 read)", 33);
    run<11>("2 x 16x16x32 i8 + 32 v_add3 (+2 reads)", 34);
    run<1>("3 x 16x16x32 i8 back to back + 32 (+3)", 35);
    run<2>("3 x 16x16x32 i8 spread + 24 (+3)", 27);
    run<4>("1 x 32x32x16 i8 + 32 (+1)", 33);
    run<5>("3 x 4x4x4 i8 + 32 (+3)", 35);
    run<12>("6 x 4x4x4 i8 + 32 (+3)", 35);
    run<6>("3 x 16x16x64 i8 + 32 (+3)", 35);
    run<40>("exec switches (2 v_add3 under them) + 32", 34);
    run<30>("32x32x16 i8, exec switches + 32 (+1 read)", 35);
    run<41>("2 LDS byte stores + 32", 32);
    run<31>("32x32x16 i8, 2 LDS byte stores + 32 (+1)", 33);
    run<42>("v_cmp -> sgpr -> v_cndmask + 32", 34);
    run<32>("32x32x16 i8, v_cmp/v_cndmask + 32 (+1)", 35);
    run<43>("4 v_mbcnt + 32", 36);
    run<33>("32x32x16 i8, 4 v_mbcnt + 32 (+1)", 37);
    run<34>("32x32x16 i8, scalar branch + 32 (+1)", 33);
    run<20>("1 x 32x32x16 i8 + 32, read 8 later", 33);
    run<21>("1 x 32x32x16 i8 + 32, read 16 later", 33);
    run<24>("1 x 32x32x16 i8 + 32, 4 reads 24 later", 36);
    run<23>("v_bfi -> 32x32x16 i8 + 32, read 24 later", 34);
    run<22>("3 x 16x16x32 i8 8 apart, read 16 later +32", 35);
    run<7>("3 x 16x16x32 i8 alone", 0);
    run<8>("1 x 32x32x16 i8 alone", 0);
    run<9>("3 x 4x4x4 i8 alone", 0);
    run<10>("3 x 16x16x64 i8 alone", 0);
    return 0;
}
