// VALU throughput per SIMD on gfx950, by instruction: W wavefronts per SIMD each run a stream of independent
// instructions (eight accumulators); cycles per instruction per SIMD = ticks * 1 / (W * instructions).
// build: hipcc --offload-arch=gfx950 -O2 -o build/ubench/valu_tput tools/ubench/valu_tput.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int kIters = 1000;
#define REP8(x) x x x x x x x x
#define I8(op) asm volatile(op(0) "\n" op(1) "\n" op(2) "\n" op(3) "\n" op(4) "\n" op(5) "\n" op(6) "\n" op(7) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c), "s"(sm), "s"(k) : "vcc");
#define OP_ADD(i)   "v_add_u32 %" #i ", %" #i ", %8"
#define OP_ADD3(i)  "v_add3_u32 %" #i ", %" #i ", %8, %9"
#define OP_SDWA(i)  "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0"
#define OP_DOT4(i)  "v_dot4_u32_u8 %" #i ", %8, %9, %" #i
#define OP_CNDV(i)  "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define OP_CNDS(i)  "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %10"
#define OP_CMPV(i)  "v_cmp_lt_u32 vcc, %" #i ", %8"
#define OP_ANDOR(i) "v_and_or_b32 %" #i ", %" #i ", %11, %8"
#define OP_LSHLOR(i) "v_lshl_or_b32 %" #i ", %" #i ", 8, %8"
#define OP_BFE(i)   "v_bfe_u32 %" #i ", %" #i ", 8, 6"
#define OP_AND(i)   "v_and_b32 %" #i ", %8, %" #i
#define OP_LSHL(i)  "v_lshlrev_b32 %" #i ", 1, %" #i
#define OP_MOV(i)   "v_mov_b32 %" #i ", %8"
#define OP_PERM(i)  "v_perm_b32 %" #i ", %" #i ", %8, %9"
#define OP_MAD24(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9"
#define OP_PKADD(i) "v_pk_add_u16 %" #i ", %" #i ", %8"
#define OP_XOR(i)   "v_xor_b32 %" #i ", %8, %" #i
#define OP_MAX3(i)  "v_max3_u32 %" #i ", %" #i ", %8, %9"
#define OP_CMPS(i)  "v_cmp_lt_u32_e64 s[20:21], %" #i ", %8"
#define OP_ADDSDWA1(i)  "v_add_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:DWORD"

template <int T>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint64_t* cyc, uint32_t seed) {
    const uint32_t lane = threadIdx.x;
    uint32_t r0 = seed + lane, r1 = seed ^ lane, r2 = lane * 3, r3 = 7, r4 = 9, r5 = 11, r6 = 13, r7 = 15;
    const uint32_t b = seed * 3 + 1, c = seed ^ 0x55;
    const uint64_t sm = 0x5555555555555555ull; const uint32_t k = 0x3F00;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
        if (T == 0) { REP8(I8(OP_ADD)) }
        if (T == 1) { REP8(I8(OP_SDWA)) }
        if (T == 2) { REP8(I8(OP_DOT4)) }
        if (T == 3) { REP8(I8(OP_CNDV)) }
        if (T == 4) { REP8(I8(OP_CNDS)) }
        if (T == 5) { REP8(I8(OP_CMPV)) }
        if (T == 6) { REP8(I8(OP_ANDOR)) }
        if (T == 7) { REP8(I8(OP_LSHLOR)) }
        if (T == 8) { REP8(I8(OP_BFE)) }
        if (T == 9) { REP8(I8(OP_AND)) }
        if (T == 10) { REP8(I8(OP_LSHL)) }
        if (T == 11) { REP8(I8(OP_PERM)) }
        if (T == 12) { REP8(I8(OP_MAD24)) }
        if (T == 13) { REP8(I8(OP_PKADD)) }
        if (T == 14) { REP8(I8(OP_ADD3)) }
        if (T == 15) { REP8(I8(OP_MAX3)) }
        if (T == 16) { REP8(asm volatile(OP_CMPS(0) "\n" OP_CMPS(1) "\n" OP_CMPS(2) "\n" OP_CMPS(3) "\n" OP_CMPS(4) "\n" OP_CMPS(5) "\n" OP_CMPS(6) "\n" OP_CMPS(7) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b) : "s20", "s21");) }
        if (T == 17) { REP8(I8(OP_ADDSDWA1)) }
        if (T == 18) { REP8(I8(OP_XOR)) }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int T> void run(const char* name) {
    printf("%-28s", name);
    for (int w : {1, 2, 4}) {
        const int blocks = 256 * 4 * w;
        uint32_t* out; uint64_t* cyc;
        hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v; s /= blocks;
        printf("  W=%d: %.2f/wave %.2f/SIMD", w, s / kIters / 64.0, s / kIters / 64.0 / w);
        hipFree(out); hipFree(cyc);
    }
    printf("\n");
}
int main() {
    printf("ticks per instruction (per wavefront; per SIMD = /W)\n");
    run<0>("v_add_u32"); run<14>("v_add3_u32"); run<1>("v_add_u32_sdwa byte,preserve"); run<17>("v_add_u32_sdwa src-sel only"); run<2>("v_dot4_u32_u8"); run<3>("v_cndmask (vcc)"); run<4>("v_cndmask_e64 (sgpr)");
    run<5>("v_cmp (vcc)"); run<16>("v_cmp_e64 (sgpr)"); run<6>("v_and_or_b32"); run<7>("v_lshl_or_b32"); run<8>("v_bfe_u32"); run<9>("v_and_b32"); run<18>("v_xor_b32"); run<10>("v_lshlrev_b32");
    run<11>("v_perm_b32"); run<12>("v_mad_u32_u24"); run<13>("v_pk_add_u16"); run<15>("v_max3_u32");
    return 0;
}
