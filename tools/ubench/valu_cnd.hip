// v_cndmask on gfx950: tools/ubench/valu_tput2.hip measured the VOP2 form reading vcc at 4-5 x the time of any other vector
// instruction.  Which part is it - the encoding, vcc as the mask, the mask's writer?  Ticks of s_memtime per instruction (pairs:
// per pair) per SIMD, W wavefronts per SIMD, eight independent accumulators.
// build: hipcc --offload-arch=gfx950 -O2 -o build/ubench/valu_cnd tools/ubench/valu_cnd.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int kIters = 1000;
#define REP8(x) x x x x x x x x
#define I8(op) asm volatile(op(0) "\n" op(1) "\n" op(2) "\n" op(3) "\n" op(4) "\n" op(5) "\n" op(6) "\n" op(7) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c), "s"(sm), "s"(k) : "vcc", "s20", "s21", "s22", "s23");
#define OP_E32(i)   "v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc"
#define OP_E64V(i)  "v_cndmask_b32_e64 %" #i ", %" #i ", %8, vcc"
#define OP_E64S(i)  "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[20:21]"
#define OP_E64K(i)  "v_cndmask_b32_e64 %" #i ", %" #i ", %8, %10"
#define OP_PAIRV(i) "v_cmp_lt_u32_e32 vcc, %" #i ", %8\n v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc"
#define OP_PAIRS(i) "v_cmp_lt_u32_e64 s[20:21], %" #i ", %8\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[20:21]"
#define OP_PAIRS2(i) "v_cmp_lt_u32_e64 s[22:23], %" #i ", %8\n s_nop 0\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, s[22:23]"
#define OP_SDWA(i)  "v_cndmask_b32_sdwa %" #i ", %" #i ", %8, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0"
#define OP_E32D(i)  "v_cndmask_b32_e32 %" #i ", %9, %8, vcc"
#define OP_ADDC(i)  "v_addc_co_u32_e32 %" #i ", vcc, %" #i ", %8, vcc"
#define OP_MOVS(i)  "v_mov_b32 %" #i ", %11"
#define OP_CMPONLY(i) "v_cmp_lt_u32_e32 vcc, %" #i ", %8"
#define OP_CMPS(i)  "v_cmp_lt_u32_e64 s[20:21], %" #i ", %8"
#define OP_CMPSDWA(i) "v_cmp_ne_u32_sdwa s[20:21], %" #i ", %8 src0_sel:BYTE_3 src1_sel:BYTE_3"
#define OP_CMPSDWAV(i) "v_cmp_ne_u32_sdwa vcc, %" #i ", %8 src0_sel:BYTE_3 src1_sel:BYTE_3"
#define OP_AND(i)   "v_and_b32 %" #i ", %8, %" #i
// when is the vcc a VOP2 v_cndmask reads "fresh"?
#define OP_DIST3(i) "v_cmp_lt_u32_e32 vcc, %" #i ", %8\n v_and_b32 %" #i ", %8, %" #i "\n v_or_b32 %" #i ", %9, %" #i "\n v_add_u32 %" #i ", %" #i ", %9\n v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc"
#define OP_THREE(i) "v_cmp_lt_u32_e32 vcc, %" #i ", %8\n v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc\n v_cndmask_b32_e32 %" #i ", %" #i ", %8, vcc\n v_cndmask_b32_e32 %" #i ", %9, %" #i ", vcc"
#define OP_SALUBETWEEN(i) "v_cmp_lt_u32_e32 vcc, %" #i ", %8\n s_mov_b64 s[22:23], exec\n s_nop 0\n v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc"
#define OP_CMPBETWEEN(i) "v_cmp_lt_u32_e32 vcc, %" #i ", %8\n v_cmp_lt_u32_e64 s[20:21], %" #i ", %9\n v_cndmask_b32_e32 %" #i ", %" #i ", %9, vcc"
#define OP_LITERAL(i) "v_cmp_gt_u32_sdwa s[20:21], %11, %" #i " src0_sel:DWORD src1_sel:WORD_0\n v_cmp_gt_u32_sdwa vcc, %11, %8 src0_sel:DWORD src1_sel:WORD_0\n v_perm_b32 %" #i ", %" #i ", %8, %9\n v_cmp_ne_u32_sdwa s[22:23], %8, %9 src0_sel:BYTE_3 src1_sel:BYTE_3\n v_cndmask_b32_e64 %" #i ", 2.0, %" #i ", s[20:21]\n v_cndmask_b32_sdwa %" #i ", %" #i ", %9, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n v_cndmask_b32_e64 %" #i ", %" #i ", 2.0, s[22:23]"
#define OP_LITERAL64(i) "v_cmp_gt_u32_sdwa s[20:21], %11, %" #i " src0_sel:DWORD src1_sel:WORD_0\n v_cmp_gt_u32_sdwa vcc, %11, %8 src0_sel:DWORD src1_sel:WORD_0\n v_perm_b32 %" #i ", %" #i ", %8, %9\n v_cmp_ne_u32_sdwa s[22:23], %8, %9 src0_sel:BYTE_3 src1_sel:BYTE_3\n v_cndmask_b32_e64 %" #i ", 2.0, %" #i ", s[20:21]\n v_cndmask_b32_e64 %" #i ", %" #i ", %9, vcc\n v_cndmask_b32_e64 %" #i ", %" #i ", 2.0, s[22:23]"

template <int T>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint64_t* cyc, uint32_t seed) {
    const uint32_t lane = threadIdx.x;
    uint32_t r0 = seed + lane, r1 = seed ^ lane, r2 = lane * 3, r3 = 7, r4 = 9, r5 = 11, r6 = 13, r7 = 15;
    const uint32_t b = seed * 3 + 1, c = seed ^ 0x55;
    const uint64_t sm = 0x5555555555555555ull; const uint32_t k = 0x3F00;
    asm volatile("s_mov_b64 vcc, %0\n s_mov_b64 s[20:21], %0" : : "s"(sm) : "vcc", "s20", "s21");
    if (T == 16) asm volatile("v_cmp_lt_u32_e32 vcc, %0, %1" : : "v"(r0), "v"(b) : "vcc");      // vcc written by a vector compare, once
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#define CASE(n, op) if (T == n) { REP8(I8(op)) }
        CASE(0, OP_E32) CASE(1, OP_E64V) CASE(2, OP_E64S) CASE(3, OP_E64K) CASE(4, OP_PAIRV) CASE(5, OP_PAIRS) CASE(6, OP_PAIRS2) CASE(7, OP_SDWA)
        CASE(8, OP_E32D) CASE(9, OP_ADDC) CASE(10, OP_MOVS) CASE(11, OP_CMPONLY) CASE(12, OP_CMPS) CASE(13, OP_CMPSDWA) CASE(14, OP_CMPSDWAV) CASE(15, OP_AND)
        CASE(16, OP_E32) CASE(17, OP_DIST3) CASE(18, OP_THREE) CASE(19, OP_SALUBETWEEN) CASE(20, OP_CMPBETWEEN) CASE(21, OP_LITERAL) CASE(22, OP_LITERAL64)
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int T> void run(const char* name) {
    printf("%-44s", name);
    for (int w : {1, 2, 4, 6}) {
        const int blocks = 256 * 4 * w;
        uint32_t* out; uint64_t* cyc;
        (void)hipMalloc(&out, blocks * 64 * 4); (void)hipMalloc(&cyc, blocks * 8);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
        (void)hipDeviceSynchronize();
        std::vector<uint64_t> h(blocks); (void)hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v; s /= blocks;
        printf("  W=%d: %.2f", w, s / kIters / 64.0 / w);
        (void)hipFree(out); (void)hipFree(cyc);
    }
    printf("\n");
}
int main() {
    printf("ticks per instruction (pairs: per pair) per SIMD\n");
    run<15>("v_and_b32 (reference, fast class)");
    run<0>("v_cndmask_b32_e32 r, r, b, vcc"); run<8>("v_cndmask_b32_e32 r, c, b, vcc (no dep on r)"); run<1>("v_cndmask_b32_e64 r, r, b, vcc"); run<2>("v_cndmask_b32_e64 r, r, b, s[20:21]");
    run<3>("v_cndmask_b32_e64 r, r, b, sgpr pair (input)"); run<7>("v_cndmask_b32_sdwa ... vcc");
    run<11>("v_cmp_lt_u32_e32 vcc"); run<12>("v_cmp_lt_u32_e64 s[20:21]"); run<13>("v_cmp_ne_u32_sdwa s[20:21]"); run<14>("v_cmp_ne_u32_sdwa vcc");
    run<4>("pair: v_cmp_e32 vcc + v_cndmask_e32 vcc"); run<5>("pair: v_cmp_e64 s + v_cndmask_e64 s"); run<6>("pair: v_cmp_e64 s, s_nop 0, v_cndmask_e64 s");
    run<9>("v_addc_co_u32 (vcc in, vcc out)"); run<10>("v_mov_b32 v, sgpr");
    run<16>("v_cndmask_e32 vcc, vcc from ONE old v_cmp");
    run<17>("v_cmp vcc, and, or, add, v_cndmask_e32 vcc (5 instr)"); run<18>("v_cmp vcc + 3 x v_cndmask_e32 vcc (4 instr)");
    run<19>("v_cmp vcc, s_mov, s_nop, v_cndmask_e32 (2 VALU)"); run<20>("v_cmp vcc, v_cmp_e64 s, v_cndmask_e32 vcc (3)");
    run<21>("literal_word's 7 instructions (sdwa cndmask vcc)"); run<22>("... with v_cndmask_e64 vcc instead (no half select)");
    return 0;
}
