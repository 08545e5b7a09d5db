// qoi_dev.h — device-side vocabulary shared by the gfx950 kernels (wave64 only).
//
// Format constants follow the reference's normative comment (qoi.h:61-206) and
// qoi.h:313-339.  Nothing here is portable: wave size 64, DPP wave_shr, v_dot4,
// LDS atomics and 64-bit ballots are used directly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "qoi_kernels.h"

namespace qoimi {

constexpr uint32_t kTagIndex = 0x00, kTagDiff = 0x40, kTagLuma = 0x80, kTagRun = 0xC0;
constexpr uint32_t kTagRgb = 0xFE, kTagRgba = 0xFF;
constexpr uint32_t kInitPx = 0xFF000000u;   // {r,g,b,a} = {0,0,0,255} as a little-endian word (qoi.h:396-399)

typedef __attribute__((address_space(1))) u64 gu64;         // global-address-space words for
typedef __attribute__((address_space(1))) unsigned gu32;    // cross-workgroup hand-off (guide G16)

// QOI_COLOR_HASH (qoi.h:322) & 63, times 4: the byte offset of the slot in a u32[64] table.
// One v_dot4_u32_u8 with the coefficients pre-multiplied by 4 (3,5,7,11 -> 12,20,28,44).
__device__ __forceinline__ uint32_t slot_byte_offset(uint32_t px) {
    return __builtin_amdgcn_udot4(px, 0x2C1C140Cu, 0u, false) & 0xFCu;
}
__host__ __device__ __forceinline__ uint32_t slot_of(uint32_t px) {
    return ((px & 0xFF) * 3u + ((px >> 8) & 0xFF) * 5u + ((px >> 16) & 0xFF) * 7u + (px >> 24) * 11u) & 63u;
}

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
// mask of the lanes where b holds (HIP's __ballot(int) first turns a bool that already sits in an SGPR pair into 0/1 in a
// VGPR and compares that with 0: two vector instructions per loop test)
__device__ __forceinline__ u64 lanes_where(bool b) { return __builtin_amdgcn_ballot_w64(b); }

// value of lane-1 (lane 0 receives `carry`): one v_mov_b32_dpp wave_shr:1.
__device__ __forceinline__ uint32_t from_lane_below(uint32_t v, uint32_t carry) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)carry, (int)v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, lane);
}
// lane index may be any wave-uniform value (SGPR)
__device__ __forceinline__ uint32_t read_lane_dyn(uint32_t v, uint32_t lane) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane);
}
__device__ __forceinline__ uint32_t gather_lane(uint32_t v, uint32_t src_lane) {
    return (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)v);
}
// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ uint32_t count_below(u64 mask) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
// base + number of set bits of `mask` strictly below this lane (base: any per-lane value)
__device__ __forceinline__ uint32_t count_below_from(u64 mask, uint32_t base) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, base));
}
__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { int t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}

// Relaxed agent-scope 8-byte granules ("the data is the flag", guide G16 R2).
__device__ __forceinline__ void granule_store(u64* p, u64 v) {
    __hip_atomic_store((gu64*)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ u64 granule_load(const u64* p) {
    return __hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace qoimi
