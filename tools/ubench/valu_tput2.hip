// More of tools/ubench/valu_tput.hip: which vector instructions run at the fast (v_add_u32 / v_and_b32) rate on gfx950 and which
// at the VOP3 rate.  W wavefronts per SIMD, eight independent accumulators each; ticks of s_memtime per instruction.
// build: hipcc --offload-arch=gfx950 -O2 -o build/ubench/valu_tput2 tools/ubench/valu_tput2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
constexpr int kIters = 1000;
#define REP8(x) x x x x x x x x
#define I8(op) asm volatile(op(0) "\n" op(1) "\n" op(2) "\n" op(3) "\n" op(4) "\n" op(5) "\n" op(6) "\n" op(7) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(b), "v"(c), "s"(sm), "s"(k) : "vcc");
#define OP_ADD(i)   "v_add_u32 %" #i ", %" #i ", %8"
#define OP_OR(i)    "v_or_b32 %" #i ", %8, %" #i
#define OP_SUB(i)   "v_sub_u32 %" #i ", %" #i ", %8"
#define OP_MOV(i)   "v_mov_b32 %" #i ", %8"
#define OP_CNDV(i)  "v_cndmask_b32 %" #i ", %" #i ", %8, vcc"
#define OP_MIN(i)   "v_min_u32 %" #i ", %" #i ", %8"
#define OP_LSHR(i)  "v_lshrrev_b32 %" #i ", 1, %" #i
#define OP_LSHL8(i) "v_lshlrev_b32 %" #i ", 8, %" #i
#define OP_NOT(i)   "v_not_b32 %" #i ", %" #i
#define OP_FFBH(i)  "v_ffbh_u32 %" #i ", %" #i
#define OP_MBCNT(i) "v_mbcnt_lo_u32_b32 %" #i ", %8, %" #i
#define OP_MUL24(i) "v_mul_u32_u24 %" #i ", %" #i ", %8"
#define OP_ALIGN(i) "v_alignbit_b32 %" #i ", %" #i ", %8, 8"
#define OP_BFI(i)   "v_bfi_b32 %" #i ", %8, %" #i ", %9"
#define OP_SUBSDWA(i) "v_sub_u32_sdwa %" #i ", %" #i ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1"
#define OP_MOVDPP(i) "v_mov_b32_dpp %" #i ", %" #i " row_shr:1 row_mask:0xf bank_mask:0xf"
#define OP_ADDDPP(i) "v_add_u32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf"
#define OP_ANDLIT(i) "v_and_b32 %" #i ", 0x7f7f7f7f, %" #i
#define OP_ADDLIT(i) "v_add_u32 %" #i ", 0x12345, %" #i
#define OP_ADDS(i)  "v_add_u32 %" #i ", %11, %" #i
#define OP_XNOR(i)  "v_xnor_b32 %" #i ", %8, %" #i
#define OP_ADDCO(i) "v_add_co_u32 %" #i ", vcc, %" #i ", %8"
#define OP_CMPEQ(i) "v_cmp_eq_u32 vcc, %" #i ", %8"
#define OP_ADD3(i)  "v_add3_u32 %" #i ", %" #i ", %8, %9"
#define OP_SAD(i)   "v_sad_u8 %" #i ", %" #i ", %8, %9"
#define OP_PKSUB(i) "v_pk_sub_u16 %" #i ", %" #i ", %8"
#define OP_MAXI(i)  "v_max_i32 %" #i ", %" #i ", %8"
#define OP_ASHR(i)  "v_ashrrev_i32 %" #i ", 3, %" #i
#define OP_BFEI(i)  "v_bfe_i32 %" #i ", %" #i ", 8, 6"
#define OP_ADDF(i)  "v_add_f32 %" #i ", %" #i ", %8"
#define OP_FMA(i)   "v_fma_f32 %" #i ", %" #i ", %8, %9"
#define OP_FMAC(i)  "v_fmac_f32 %" #i ", %8, %9"
#define OP_CVT(i)   "v_cvt_f32_ubyte0 %" #i ", %" #i
#define OP_ADD16(i) "v_add_u16 %" #i ", %" #i ", %8"

template <int T>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint64_t* cyc, uint32_t seed) {
    const uint32_t lane = threadIdx.x;
    uint32_t r0 = seed + lane, r1 = seed ^ lane, r2 = lane * 3, r3 = 7, r4 = 9, r5 = 11, r6 = 13, r7 = 15;
    const uint32_t b = seed * 3 + 1, c = seed ^ 0x55;
    const uint64_t sm = 0x5555555555555555ull; const uint32_t k = 0x3F00;
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
#define CASE(n, op) if (T == n) { REP8(I8(op)) }
        CASE(0, OP_ADD) CASE(1, OP_OR) CASE(2, OP_SUB) CASE(3, OP_MOV) CASE(4, OP_CNDV) CASE(5, OP_MIN) CASE(6, OP_LSHR) CASE(7, OP_LSHL8)
        CASE(8, OP_NOT) CASE(9, OP_FFBH) CASE(10, OP_MBCNT) CASE(11, OP_MUL24) CASE(12, OP_ALIGN) CASE(13, OP_BFI) CASE(14, OP_SUBSDWA)
        CASE(15, OP_MOVDPP) CASE(16, OP_ADDDPP) CASE(17, OP_ANDLIT) CASE(18, OP_ADDLIT) CASE(19, OP_ADDS) CASE(20, OP_XNOR) CASE(21, OP_ADDCO)
        CASE(22, OP_CMPEQ) CASE(23, OP_ADD3) CASE(24, OP_SAD) CASE(25, OP_PKSUB) CASE(26, OP_MAXI) CASE(27, OP_ASHR) CASE(28, OP_BFEI)
        CASE(29, OP_ADDF) CASE(30, OP_FMA) CASE(31, OP_FMAC) CASE(32, OP_CVT) CASE(33, OP_ADD16)
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int T> void run(const char* name) {
    printf("%-30s", name);
    for (int w : {1, 2, 4, 6}) {
        const int blocks = 256 * 4 * w;
        uint32_t* out; uint64_t* cyc;
        hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
        hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
        hipDeviceSynchronize();
        std::vector<uint64_t> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
        double s = 0; for (auto v : h) s += (double)v; s /= blocks;
        printf("  W=%d: %.2f/SIMD", w, s / kIters / 64.0 / w);
        hipFree(out); hipFree(cyc);
    }
    printf("\n");
}
int main() {
    printf("ticks of s_memtime per instruction per SIMD (W wavefronts per SIMD)\n");
    run<0>("v_add_u32"); run<1>("v_or_b32"); run<2>("v_sub_u32"); run<3>("v_mov_b32"); run<4>("v_cndmask_b32 (vcc, e32)"); run<5>("v_min_u32"); run<26>("v_max_i32");
    run<6>("v_lshrrev_b32"); run<7>("v_lshlrev_b32 8"); run<27>("v_ashrrev_i32"); run<8>("v_not_b32"); run<9>("v_ffbh_u32"); run<10>("v_mbcnt_lo_u32_b32"); run<11>("v_mul_u32_u24"); run<12>("v_alignbit_b32");
    run<13>("v_bfi_b32"); run<28>("v_bfe_i32"); run<14>("v_sub_u32_sdwa src-sel"); run<15>("v_mov_b32_dpp row_shr:1"); run<16>("v_add_u32_dpp row_shr:1"); run<17>("v_and_b32 literal"); run<18>("v_add_u32 literal");
    run<19>("v_add_u32 sgpr operand"); run<20>("v_xnor_b32"); run<21>("v_add_co_u32 (vcc)"); run<22>("v_cmp_eq_u32 (vcc)"); run<23>("v_add3_u32"); run<24>("v_sad_u8"); run<25>("v_pk_sub_u16");
    run<33>("v_add_u16"); run<29>("v_add_f32"); run<30>("v_fma_f32"); run<31>("v_fmac_f32"); run<32>("v_cvt_f32_ubyte0");
    return 0;
}
