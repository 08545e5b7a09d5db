// Instruction cost on gfx950 as ONE wavefront sees it (and with a second wavefront on the same SIMD): cycles per
// instruction for dependent chains and for independent streams of the operations the QOI kernels are made of.
// build: hipcc --offload-arch=gfx950 -O2 -o build/ubench/valu_lat tools/ubench/valu_lat.hip ; run: build/ubench/valu_lat
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP16(x) x x x x x x x x x x x x x x x x
constexpr int kIters = 2000;

template <int T>
__global__ __launch_bounds__(64) void k(uint32_t* out, uint64_t* cyc, uint32_t seed) {
    __shared__ uint32_t lds[64 * 64];
    const uint32_t lane = threadIdx.x;
    for (int i = 0; i < 64; ++i) lds[i * 64 + lane] = (i * 7 + lane) & 63;
    __syncthreads();
    uint32_t a = seed + lane, b = seed * 3 + 1, c = seed ^ 0x55, d = lane, e = 5, f = 6, g = 7, h = 8;
    const uint32_t base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint32_t*)&lds[lane];
    const uint64_t t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
        if (T == 0) { REP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));) }
        if (T == 1) { REP16(asm volatile("v_add_u32_sdwa %0, %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0" : "+v"(a) : "v"(b));) }
        if (T == 2) { REP16(asm volatile("v_dot4_u32_u8 %0, %1, %2, %0" : "+v"(a) : "v"(b), "v"(c));) }
        if (T == 3) { REP16(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));) }
        if (T == 4) { REP16(asm volatile("v_add_u32_sdwa %0, %0, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n v_add_u32_sdwa %1, %1, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n v_add_u32_sdwa %2, %2, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0\n v_add_u32_sdwa %3, %3, %4 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b));) }
        if (T == 5) { REP16(asm volatile("v_dot4_u32_u8 %0, %4, %5, %0\n v_dot4_u32_u8 %1, %4, %5, %1\n v_dot4_u32_u8 %2, %4, %5, %2\n v_dot4_u32_u8 %3, %4, %5, %3" : "+v"(a), "+v"(d), "+v"(e), "+v"(f) : "v"(b), "v"(c));) }
        if (T == 6) { REP16(asm volatile("v_lshlrev_b32 %1, 8, %0\n v_and_or_b32 %1, %1, %3, %2\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(d) : "v"(base), "s"(0x3F00));) }
        if (T == 7) { REP16(asm volatile("v_cmp_gt_i32 vcc, 0, %0\n s_cbranch_vccnz 0\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b) : "vcc");) }
        if (T == 8) { REP16(asm volatile("v_cmp_lt_u32 vcc, %2, %0\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a) : "v"(b), "v"(c) : "vcc");) }
        if (T == 9) { REP16(asm volatile("v_lshlrev_b32 %1, 8, %0\n v_and_or_b32 %1, %1, %3, %2\n ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(d) : "v"(base), "s"(0x3F00));) }
        if (T == 10) { REP16(asm volatile("s_add_u32 %0, %0, 1" : "+s"(g) : : "scc");) }
        if (T == 11) { REP16(asm volatile("v_cmp_gt_i32 vcc, 0, %0\n s_and_b64 vcc, exec, vcc\n s_cbranch_vccnz 0\n v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b) : "vcc", "scc");) }
        if (T == 12) { REP16(asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (T == 13) { REP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a) : "v"(b), "v"(c));) }
        if (T == 14) { REP16(asm volatile("v_add_u32 %0, %0, %1\n s_nop 0" : "+v"(a) : "v"(b));) }
    }
    const uint64_t t1 = __builtin_readcyclecounter();
    out[blockIdx.x * 64 + lane] = a + d + e + f + g + h;
    if (lane == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int T> void run(const char* name, int per_rep, int waves_per_simd) {
    const int blocks = 256 * 4 * waves_per_simd;     // 64 KiB... occupancy by grid size only: 16 KiB LDS per block
    uint32_t* out; uint64_t* cyc;
    hipMalloc(&out, blocks * 64 * 4); hipMalloc(&cyc, blocks * 8);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<T>, dim3(blocks), dim3(64), 0, 0, out, cyc, 12345u);
    hipEventRecord(e1, 0);
    hipDeviceSynchronize();
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    std::vector<uint64_t> h(blocks); hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v; s /= blocks;
    // readcyclecounter = s_memtime: constant 100 MHz clock on gfx9?  report raw ticks per instruction too
    printf("%-34s waves/SIMD %d  ticks/instr %.3f   kernel %.3f ms = %.2f ns/instr\n", name, waves_per_simd, s / kIters / (16.0 * per_rep), ms, ms * 1e6 / kIters / (16.0 * per_rep));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int w : {1, 2}) {
        run<0>("v_add_u32 dependent", 1, w);
        run<1>("v_add_u32_sdwa dependent", 1, w);
        run<2>("v_dot4_u32_u8 dependent", 1, w);
        run<3>("v_add_u32 x4 independent", 4, w);
        run<4>("v_add_u32_sdwa x4 independent", 4, w);
        run<5>("v_dot4_u32_u8 x4 independent", 4, w);
        run<6>("lshl,and_or,ds_read,wait chain", 4, w);
        run<7>("v_cmp,s_cbranch_vccnz(nt),v_add", 3, w);
        run<8>("v_cmp,v_cndmask dependent", 2, w);
        run<9>("lshl,and_or,ds_write,ds_read,wait", 5, w);
        run<10>("s_add_u32 dependent", 1, w);
        run<11>("v_cmp,s_and,s_cbranch(nt),v_add", 4, w);
        run<12>("v_perm_b32 dependent", 1, w);
        run<13>("v_mad_u32_u24 dependent", 1, w);
        run<14>("v_add_u32,s_nop dependent", 2, w);
    }
    return 0;
}
