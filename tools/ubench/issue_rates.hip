// Micro-benchmark (diagnostic, not product): issue rates of VALU / SALU / DS byte writes / ds_wrxchg
// on gfx950 as a function of waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 issue_rates.hip -o issue_rates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
typedef unsigned long long u64;

#define REP16(x) x x x x x x x x x x x x x x x x

template <int MODE>
__global__ __launch_bounds__(256) void kern(uint32_t* out, int iters, uint32_t seed) {
    __shared__ uint32_t lds[4][1024];
    uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t a = seed + threadIdx.x, b = seed * 3 + 1, c = a ^ b, d = a + 7;
    uint32_t s0 = __builtin_amdgcn_readfirstlane(seed), s1 = s0 + 5, s2 = s0 ^ 9, s3 = s0 * 3;
    lds[wave][lane] = a;
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {        // 64 independent-ish VALU
            REP16(asm volatile("v_add_u32 %0, %0, %1\n v_xor_b32 %2, %2, %3\n v_add_u32 %1, %1, %2\n v_xor_b32 %3, %3, %0" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));)
        } else if (MODE == 1) { // 64 SALU
            REP16(asm volatile("s_add_u32 %0, %0, %1\n s_xor_b32 %2, %2, %3\n s_add_u32 %1, %1, %2\n s_xor_b32 %3, %3, %0" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");)
        } else if (MODE == 2) { // 64 VALU + 64 SALU interleaved
            REP16(asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_xor_b32 %2, %2, %3\n s_xor_b32 %6, %6, %7\n v_add_u32 %1, %1, %2\n s_add_u32 %5, %5, %6\n v_xor_b32 %3, %3, %0\n s_xor_b32 %7, %7, %4"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");)
        } else if (MODE == 3) { // 64 VALU + 32 SALU
            REP16(asm volatile("v_add_u32 %0, %0, %1\n s_add_u32 %4, %4, %5\n v_xor_b32 %2, %2, %3\n v_add_u32 %1, %1, %2\n s_xor_b32 %6, %6, %7\n v_xor_b32 %3, %3, %0"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) :: "scc");)
        } else if (MODE == 4) { // 16 ds_write_b8, distinct dense addresses
            uint32_t addr = (uint32_t)(uintptr_t)&lds[wave][0] + lane + (i & 7);
            REP16(asm volatile("ds_write_b8 %0, %1" :: "v"(addr), "v"(a) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 5) { // 16 ds_wrxchg_rtn_b32 on a 64-entry table, pseudo-random slots
            uint32_t addr = (uint32_t)(uintptr_t)&lds[wave][0] + ((a * 2654435761u >> 26) << 2);
            REP16(asm volatile("ds_wrxchg_rtn_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)" : "+v"(a) : "v"(addr) : "memory");)
        } else if (MODE == 6) { // 16 ds_wrxchg, no wait between (different data regs)
            uint32_t addr = (uint32_t)(uintptr_t)&lds[wave][0] + ((a * 2654435761u >> 26) << 2);
            REP16(asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2" : "=v"(b) : "v"(addr), "v"(c) : "memory");)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else if (MODE == 7) { // 64 VALU with SGPR-mask exec changes: s_mov exec + v_op pairs
            u64 m = 0xFFFF0000FFFF0000ull | s0;
            REP16(asm volatile("s_mov_b64 exec, %4\n v_add_u32 %0, %0, %1\n s_mov_b64 exec, -1\n v_xor_b32 %2, %2, %3\n v_add_u32 %1, %1, %2\n v_xor_b32 %3, %3, %0"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(m));)
        } else if (MODE == 8) { // v_cmp -> sgpr + v_cndmask with sgpr mask pairs (32 + 32)
            u64 m;
            REP16(asm volatile("v_cmp_lt_u32 %4, %0, %1\n v_cndmask_b32 %2, %2, %3, %4\n v_cmp_gt_u32 %4, %2, %3\n v_cndmask_b32 %0, %0, %1, %4"
                               : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "=&s"(m));)
        } else if (MODE == 9) { // mbcnt pairs
            u64 m = 0xF0F0F0F0F0F0F0F0ull ^ s0;
            REP16(asm volatile("v_mbcnt_lo_u32_b32 %0, %2, %0\n v_mbcnt_hi_u32_b32 %0, %3, %0\n v_mbcnt_lo_u32_b32 %1, %2, %1\n v_mbcnt_hi_u32_b32 %1, %3, %1"
                               : "+v"(a), "+v"(b) : "s"((uint32_t)m), "s"((uint32_t)(m >> 32)));)
        } else if (MODE == 10) { // dpp movs + readlane
            REP16(asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_readlane_b32 %2, %1, 63\n v_mov_b32_dpp %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf\n v_readlane_b32 %3, %0, 63"
                               : "+v"(a), "+v"(b), "=s"(s0), "=s"(s1));)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + s0 + s1 + s2 + s3 + lds[wave][lane];
}

template <int MODE>
static void run(const char* name, int ops_per_iter, uint32_t* d) {
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {           // waves per SIMD: blocks of 256 threads = 1 wave per SIMD each
        int blocks = 256 * wps;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        kern<MODE><<<blocks, 256>>>(d, 10, 1);
        hipEventRecord(e0);
        kern<MODE><<<blocks, 256>>>(d, iters, 1);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double per_simd_ops = (double)iters * ops_per_iter * wps;      // wave-instructions issued per SIMD
        printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (%.2f cyc @2.4GHz)\n", name, wps, ms,
               ms * 1e6 / per_simd_ops, ms * 1e6 / per_simd_ops * 2.4);
    }
}

int main() {
    uint32_t* d; hipMalloc(&d, 256 * 8 * 256 * 4);
    run<0>("valu x64", 64, d);
    run<1>("salu x64", 64, d);
    run<2>("valu64+salu64 (per 128)", 128, d);
    run<3>("valu64+salu32 (per 96)", 96, d);
    run<4>("ds_write_b8 x16", 16, d);
    run<5>("ds_wrxchg_rtn dep x16", 16, d);
    run<6>("ds_wrxchg_rtn indep x16", 16, d);
    run<7>("valu64+exec moves32 (per 96)", 96, d);
    run<8>("v_cmp+cndmask x64", 64, d);
    run<9>("mbcnt x64", 64, d);
    run<10>("dpp+readlane x64", 64, d);
    hipFree(d);
    return 0;
}
