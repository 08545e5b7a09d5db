// qoi_encode.hip — MI355X-native QOI encoder kernels (gfx950, wave64).
//
// Replaces the sequential loop of the reference encoder (qoi.h:356-486) with a
// slab-parallel evaluation of the SAME function of the input pixels; the emitted
// stream is byte-identical to the reference's (tests/test_gpu_parity.py).
//
// Every output byte is a pure function of the input:
//   * prev pixel           = the neighbouring input pixel (qoi.h:477)
//   * "edge" pixel         = px[i] != px[i-1] (qoi.h:415)
//   * run bytes            = function of the distance to the last edge (qoi.h:416-428)
//   * colour-table content = last edge pixel per hash slot (qoi.h:430-436)
// so an image is cut into SETS of R slabs of 1024 pixels (one wavefront per set) and only
// three small quantities are carried between sets:
//   (1) the 64-entry colour table       -> replay of the 512 pixels before the set (hot path),
//                                          passes E1 (slab summary) + E2 (scan) for flat content
//   (2) the position of the last edge   -> same
//   (3) the output byte offset          -> decoupled look-back inside pass E3
//
// Pass E3 (enc_sets) is the hot kernel: two coalesced dword loads per pixel (the pixel and the
// one before it), an LDS-resident colour table per wavefront, 64-bit ballots / mbcnt for run
// lengths and byte offsets.  Byte/integer work only - no MFMA.
#include <stdlib.h>
#include "qoi_dev.h"
#include "qoi_kernels.h"

namespace qoimi {

#ifdef QOIMI_ENC_PHASES
// Diagnostic build (-DQOIMI_ENC_PHASES, tools/measure/enc_phases.py): where a wavefront of enc_sets spends its life - s_memtime
// ticks per phase of a set, summed over all wavefronts.  [0] entry state, [1] the groups inside the image, [2] groups of the general
// form, [3] look-back, [4] copy-out, [5] sets, [6] sets whose first poll (asked for a group ahead) did not suffice, [7] re-polls.
__device__ unsigned long long g_enc_phase[8];
#define PHASE_MARK(k) do { const unsigned long long t_now = __builtin_readcyclecounter(); if (lane == 0) atomicAdd(&g_enc_phase[k], t_now - t_mark); t_mark = t_now; } while (0)
#else
#define PHASE_MARK(k) do { } while (0)
#endif

// ---------------------------------------------------------------------------------
// pixel load: CH = 4 -> one dword; CH = 3 -> three bytes, alpha forced to 255
// (the reference leaves alpha at its 255 start value for 3-channel input, qoi.h:399-413)
// ---------------------------------------------------------------------------------
template <int CH>
__device__ __forceinline__ uint32_t load_px(const uint8_t* __restrict__ img, uint32_t i) {
    if constexpr (CH == 4) {
        return reinterpret_cast<const uint32_t*>(img)[i];
    } else {
        const uint8_t* p = img + (size_t)i * 3u;
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xFF000000u;
    }
}

// the same relative to a per-lane pointer: pixel `k` (a compile-time constant, may be negative) after the one q points to
template <int CH>
__device__ __forceinline__ uint32_t load_px_at(const uint8_t* __restrict__ q, int k) {
    if constexpr (CH == 4) {
        return reinterpret_cast<const uint32_t*>(q)[k];
    } else {
        const uint8_t* p = q + k * 3;
        return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | 0xFF000000u;
    }
}

__device__ __forceinline__ int msb64(u64 m) { return 63 - __builtin_clzll(m); }

// The geometry of image `img`: the call's one shape (MIXED false: the same expressions as ever, folded by the compiler), or its entry
// of the image table (qoimi_encode_images).
template <bool MIXED>
__device__ __forceinline__ EncImage enc_image(const EncParams& p, uint32_t img) {
    if (MIXED) return p.img_tab[img];
    EncImage e;
    e.pixel_off = (size_t)img * p.pixel_stride; e.out_off = (size_t)img * p.out_stride;
    e.npx = p.npx; e.spi = p.spi; e.gpi = p.gpi; e.sets = p.sets_per_image;
    e.set_base = img * p.sets_per_image; e.slab_base = img * p.spi; e.grp_base = img * p.gpi; e.unit_base = 0u;
    e.width = p.width; e.height = p.height; e.colorspace = p.colorspace; e.len_index = img;
    return e;
}
// image that holds global slab / group / set / unit `v` (MIXED): the last image whose base is <= v
#define QOIMI_FIND_ENC_IMAGE(FIELD)                                                                    \
    __device__ __forceinline__ uint32_t find_by_##FIELD(const EncParams& p, uint32_t v) {             \
        uint32_t lo = 0, hi = p.n_images;                                                              \
        while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (p.img_tab[mid].FIELD <= v) lo = mid; else hi = mid; } \
        return lo;                                                                                     \
    }
QOIMI_FIND_ENC_IMAGE(slab_base)
QOIMI_FIND_ENC_IMAGE(grp_base)
QOIMI_FIND_ENC_IMAGE(set_base)
QOIMI_FIND_ENC_IMAGE(unit_base)

// sum of v over lanes 0..stop (stop >= 63: all lanes), the same value for every lane.  Seven DPP adds: within the rows of 16
// lanes (row_shr 1, 2, 3, then 4 and 8 on the banks that have such a neighbour), then row_bcast 15 / 31 carry the row totals
// upwards; lane 63 holds the total.
__device__ __forceinline__ uint32_t wave_sum32_upto(uint32_t v, uint32_t lane, int stop) {
    const int v0 = (int)lane <= stop ? (int)v : 0;
    int s = v0;
    s += __builtin_amdgcn_update_dpp(0, v0, 0x111, 0xf, 0xf, true);      // row_shr:1
    s += __builtin_amdgcn_update_dpp(0, v0, 0x112, 0xf, 0xf, true);      // row_shr:2
    s += __builtin_amdgcn_update_dpp(0, v0, 0x113, 0xf, 0xf, true);      // row_shr:3   -> sums of 4
    s += __builtin_amdgcn_update_dpp(0, s, 0x114, 0xf, 0xe, true);       // row_shr:4, banks 1..3 -> sums of 8
    s += __builtin_amdgcn_update_dpp(0, s, 0x118, 0xf, 0xc, true);       // row_shr:8, banks 2..3 -> lane 15 of a row: the row's sum
    s += __builtin_amdgcn_update_dpp(0, s, 0x142, 0xa, 0xf, true);       // row_bcast:15 into rows 1 and 3
    s += __builtin_amdgcn_update_dpp(0, s, 0x143, 0xc, 0xf, true);       // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane(s, 63);
}

// ---------------------------------------------------------------------------------
// E1: per-slab summary.  For each hash slot the LAST edge pixel of the slab (+ valid
// bit) and the position of the slab's last edge (-1: none).   [qoi.h:415,430,436]
// One wavefront per slab; LDS ds_max_u64 on (position,value) keys keeps the latest.
// ---------------------------------------------------------------------------------
template <int CH, int K, bool MIXED>
__global__ __launch_bounds__(256) void enc_slab_summary(EncParams p) {
    __shared__ u64 s_key[4][64];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    if (p.only_flagged && *p.any_generic == 0u) return;
    const uint32_t total = p.total_slabs;
#pragma unroll 1
    for (uint32_t blk = blockIdx.x; blk * 4u < total; blk += gridDim.x) {
    const uint32_t g = blk * 4u + wave;
    if (g >= total) continue;
    const uint32_t img = MIXED ? find_by_slab_base(p, g) : g / p.spi;
    const EncImage I = enc_image<MIXED>(p, img);
    const uint32_t s = g - I.slab_base;
    if (p.only_flagged && p.need_generic[img] == 0u) continue;
    const uint8_t* __restrict__ pix = p.pixels + I.pixel_off;
    const uint32_t n = I.npx, lo = s * (64u * K);

    s_key[wave][lane] = 0;
    // all K loads of the slab in flight at once (a streaming kernel with 4 loads per lane in flight
    // is bound by HBM latency, ~3 TB/s)
    uint32_t cur[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const uint32_t i = lo + t * 64u + lane;
        cur[t] = i < n ? load_px<CH>(pix, i) : 0u;
    }
    uint32_t carry = (lo > 0) ? load_px<CH>(pix, lo - 1) : kInitPx;
    int le = -1;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const uint32_t i = lo + t * 64u + lane;
        const bool inb = i < n;
        const uint32_t px = cur[t];
        const uint32_t prev = from_lane_below(px, carry);
        const bool edge = inb && px != prev;
        if (edge) {
            const u64 key = ((u64)(t * 64u + lane + 1u) << 32) | px;
            atomicMax(&s_key[wave][slot_byte_offset(px) >> 2], key);
        }
        const u64 E = __ballot(edge);
        if (E) le = (int)(lo + t * 64u) + msb64(E);
        carry = read_lane(px, 63);
    }
    __builtin_amdgcn_wave_barrier();
    const u64 k = s_key[wave][lane];
    const u64 vmask = __ballot(k != 0);
    p.sum_tab[(size_t)g * 64u + lane] = (uint32_t)k;
    if (lane == 0) { p.sum_valid[g] = vmask; p.sum_le[g] = le; }
    __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------
// E2a: exclusive "latest valid per slot" / max scan over the <=64 slabs of one group.
// lane = hash slot.  Writes per-slab group-local entry state and the group aggregate.
// ---------------------------------------------------------------------------------
template <bool MIXED>
__global__ __launch_bounds__(64) void enc_scan_groups(EncParams p) {
    const uint32_t lane = lane_id();
    const uint32_t G = blockIdx.x;                       // img * gpi + grp
    const uint32_t img = MIXED ? find_by_grp_base(p, G) : G / p.gpi;
    const EncImage I = enc_image<MIXED>(p, img);
    const uint32_t grp = G - I.grp_base;
    if (p.only_flagged && p.need_generic[img] == 0u) return;
    const uint32_t s0 = grp * 64u;
    const uint32_t s1 = min(I.spi, s0 + 64u);
    uint32_t cur = 0; bool curv = false; int curle = -1;
    for (uint32_t s = s0; s < s1; ++s) {
        const size_t g = (size_t)I.slab_base + s;
        const uint32_t t = p.sum_tab[g * 64u + lane];
        const u64 vm = p.sum_valid[g];
        const int l = p.sum_le[g];
        p.ent_tab[g * 64u + lane] = cur;
        const u64 cv = __ballot(curv);
        if (lane == 0) { p.ent_valid[g] = cv; p.ent_le[g] = curle; }
        if ((vm >> lane) & 1ull) { cur = t; curv = true; }
        curle = max(curle, l);
    }
    p.grp_tab[(size_t)G * 64u + lane] = cur;
    const u64 cv = __ballot(curv);
    if (lane == 0) { p.grp_valid[G] = cv; p.grp_le[G] = curle; }
}

// E2b: exclusive scan over the groups of one image (one wavefront per image).
template <bool MIXED>
__global__ __launch_bounds__(64) void enc_scan_images(EncParams p) {
    const uint32_t lane = lane_id();
    const uint32_t img = blockIdx.x;
    if (p.only_flagged && p.need_generic[img] == 0u) return;
    const EncImage I = enc_image<MIXED>(p, img);
    uint32_t cur = 0; int curle = -1;          // table starts zeroed (qoi.h:393), no edge yet
    for (uint32_t gr = 0; gr < I.gpi; ++gr) {
        const size_t G = (size_t)I.grp_base + gr;
        const uint32_t t = p.grp_tab[G * 64u + lane];
        const u64 vm = p.grp_valid[G];
        const int l = p.grp_le[G];
        p.gent_tab[G * 64u + lane] = cur;
        if (lane == 0) p.gent_le[G] = curle;
        if ((vm >> lane) & 1ull) cur = t;
        curle = max(curle, l);
    }
}

// ---------------------------------------------------------------------------------
// E3: the hot kernel.  One wavefront encodes a SET of R consecutive slabs of one image
// (R = p.set_slabs, 1..8): the colour table, the distance to the last edge and the staged
// bytes carry over from slab to slab, so the entry state (below) is paid once per set and
// the set's bytes leave the LDS in ONE piece.
//
// PROBE selects how the colour table is probed/updated for the 64 pixels of a step:
//   0  ds_or_b64 lane masks + ds_bpermute (order-independent, always valid)
//   1  one ds_wrxchg_rtn_b32 per step: relies on the LDS serving the lanes of one
//      instruction that hit the same address in ascending lane order - MEASURED at context
//      creation by lds_order_selftest; the host only picks 1 when that test passes.
// GEN: the general form of a step - lanes beyond the image's last pixel are masked, the last
//      pixel closes its run, the probe is restricted to the edge lanes.  Used for the image's
//      FIRST set and for the group of steps that holds the image's last pixel; every other
//      step runs the plain form (all 64 lanes valid, see probe_swap_all).
//
// Instruction budget (profiles/r03_s3_sq_counters_encode.txt): a wave64 vector instruction
// occupies its SIMD for 4 cycles whatever it is, scalar instructions ride along almost for free (a third of a vector
// instruction's cost at most), a DS op takes 4 LDS cycles of the CU whatever its width.  The kernel is bound by its
// vector instruction count (49 per 64 pixels), so the step below is written for few of them:
//   * every pixel emits at most ONE chunk: a repeat pixel carries the run byte of the run it
//     closes (qoi.h:417-421,425-428 put that byte in front of the next edge's chunk, which is
//     the same stream position), so chunk length is in {0,1,2,4,5} per lane;
//   * the previous pixel of every lane is LOADED with the pixel (one 8-byte request per lane, 4 bytes - 3 for 3-channel
//     input - in front of the pixel's address) instead of moved across lanes;
//   * per-lane predicates live as 64-bit lane masks in SGPRs (ballot results); the few mask
//     combinations are explicit scalar ops and come back as exec / v_cndmask masks through
//     inverse_ballot;
//   * chunk words keep byte 0 in bits 0..7 and byte 1 in bits 16..23 (ds_write_b8 /
//     ds_write_b8_d16_hi take them from there); two-byte words are the negative ones, 0x40000000 marks a long chunk;
//   * the DIFF / LUMA classification of an edge pixel is worked out for two consecutive steps at once in the 16-bit
//     halves of the registers (classify_pair; byte/integer work only - a matrix-pipe form of it was measured in round 3 and
//     removed: 19 % fewer vector instructions, no faster, EXPERIMENTS.md);
//   * work that a step does not need is skipped by wave-uniform branches (no edges: no
//     hash/probe/deltas; no 4/5-byte chunk: no third offset count).
// Chunk bytes go to a per-wave LDS staging buffer; when the set is done its byte offset in the
// stream is found by decoupled look-back over the earlier sets of the image and the staged bytes
// are copied out with aligned 16-byte stores.  A set whose bytes do not fit the staging buffer
// (more than ~1.4 bytes per pixel) spills whole 16-byte pieces to its scratch slot as it goes and
// moves them to their place itself after the look-back.
constexpr int kGroupSteps = 8;                          // steps whose pixels are loaded together (one register group)
constexpr uint32_t kGroupPx = 64u * kGroupSteps;
#ifndef QOIMI_ENC_STAGE_BYTES
#define QOIMI_ENC_STAGE_BYTES 6336
#endif
#ifndef QOIMI_ENC_WAVES_PER_SIMD
#define QOIMI_ENC_WAVES_PER_SIMD 6
#endif
constexpr uint32_t kStageBytes = QOIMI_ENC_STAGE_BYTES; // staging buffer of a wavefront (6 workgroups of 4 per CU: 4 x 6656 x 6 = 156 KB of LDS)

template <int PROBE, uint32_t STAGE>
struct EncLds {
    static constexpr uint32_t kStageDwords = STAGE / 4u + 8u;
    static constexpr uint32_t kSpill = STAGE - kGroupSteps * 320u - 16u;   // more staged bytes than this before a group: spill first
    alignas(256) uint32_t table[64];   // 256-byte aligned: slot address = base | (4*slot)
    alignas(16) uint32_t stage[kStageDwords];
    u64 mask[PROBE == 0 ? 64 : 1];     // PROBE 0 only
};

template <int PROBE> using EncLdsFor = EncLds<PROBE, kStageBytes>;

typedef __attribute__((address_space(3))) uint8_t lds_u8;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
__device__ __forceinline__ uint32_t lds_addr(const void* p) { return (uint32_t)(uintptr_t)(const lds_u8*)p; }

__device__ __forceinline__ bool in_mask(u64 m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
// v_ffbh_u32: number of leading zeros, 0xFFFFFFFF for 0
__device__ __forceinline__ uint32_t ffbh(uint32_t v) {
    uint32_t r;
    asm("v_ffbh_u32 %0, %1" : "=v"(r) : "v"(v));
    return r;
}
// lane l: v of lane l-1; lane 0: `before` of lane 63 (the previous 64 pixels).  Two DPP moves,
// no trip through an SGPR.  (Entry-state replay only; the steps load their previous pixels.)
__device__ __forceinline__ uint32_t prev_pixels(uint32_t v, uint32_t before) {
    const int first = __builtin_amdgcn_mov_dpp((int)before, 0x13C, 0xf, 0xf, false);              // wave_ror:1 (every lane has a source)
    return (uint32_t)__builtin_amdgcn_update_dpp(first, (int)v, 0x138, 0xf, 0xf, false);          // wave_shr:1, lane 0 keeps `first`
}

// a lane mask the compiler may have lost track of as wave-uniform -> SGPR pair (free when it already is one)
__device__ __forceinline__ u64 uniform64(u64 m) {
    // (the builtin returns int: without the casts to uint32_t a low half with bit 31 set sign-extends over the high half - the
    // generic path then took 32 slots of the image-level table for group-local ones whenever slot 31 had been written in the group)
    return (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m) | ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32);
}
// 64 + number of leading zeros of a non-zero lane mask, on the scalar unit (the result stays in an SGPR)
__device__ __forceinline__ uint32_t clz64_plus64(u64 m) {
    uint32_t r;
    asm("s_flbit_i32_b64 %0, %1\n\ts_or_b32 %0, %0, 64" : "=s"(r) : "s"(m) : "scc");
    return r;
}
// v + 64 on the scalar unit (v is wave-uniform; the readfirstlane is free when the compiler already holds it in an SGPR)
__device__ __forceinline__ uint32_t scalar_add64(uint32_t v) {
    uint32_t r;
    asm("s_add_u32 %0, %1, 64" : "=s"(r) : "s"(__builtin_amdgcn_readfirstlane((int)v)) : "scc");
    return r;
}
// Byte 0 (bits 0..7 of w) of every chunk and byte 1 (bits 16..23) of the 2-byte chunks of one step into the LDS
// staging buffer.  Runs with all 64 lanes enabled (the step loop is wave-uniform), so exec is switched with plain
// moves.  LDS ops of a wave complete in issue order, so the later (compiler-visible) reads of the staging buffer
// see these stores.
__device__ __forceinline__ void stage_short(uint32_t addr, uint32_t w, u64 any, u64 second) {
    // (the three exec moves cost little: writing with all lanes instead - wrong bytes, timing only - made the kernel 1-2 % faster;
    // the kernel is bound by its VECTOR instruction count, profiles/r03_s2_sq_counters_encode.txt)
    asm volatile("s_mov_b64 exec, %2\n\t"
                 "ds_write_b8 %0, %1\n\t"
                 "s_mov_b64 exec, %3\n\t"
                 "ds_write_b8_d16_hi %0, %1 offset:1\n\t"
                 "s_mov_b64 exec, -1"
                 : : "v"(addr), "v"(w), "s"(any), "s"(second) : "memory");
}
// bytes 2, 3 (g, b) of the 4/5-byte chunks and byte 4 (a) of the 5-byte chunks (exec switching instead of divergent
// branches: a divergent branch anywhere in the step makes the compiler restructure its wave-uniform branches as well)
__device__ __forceinline__ void stage_long_rest(uint32_t addr, uint32_t px, u64 lng, u64 five) {
    uint32_t t;
    asm volatile("s_mov_b64 exec, %3\n\t"
                 "v_lshrrev_b32 %0, 8, %2\n\t"
                 "ds_write_b8 %1, %0 offset:2\n\t"
                 "ds_write_b8_d16_hi %1, %2 offset:3\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "ds_write_b8_d16_hi %1, %0 offset:4\n\t"
                 "s_mov_b64 exec, -1"
                 : "=&v"(t) : "v"(addr), "v"(px), "s"(lng), "s"(five) : "memory");
}
// Colour-table probe of one step (PROBE 1): the edge lanes swap their pixel into their slot and get
// what the slot held.  Lanes of one instruction that hit the same slot are served in ascending lane
// order (lds_order_selftest), i.e. in pixel order - exactly the sequential probe/update of
// qoi.h:430-436.  The result is only defined for the lanes in `edges`; wait with probe_wait().
__device__ __forceinline__ uint32_t probe_swap(uint32_t addr, uint32_t px, u64 edges) {
    uint32_t seen;
    asm volatile("s_mov_b64 exec, %3\n\t"
                 "ds_wrxchg_rtn_b32 %0, %1, %2\n\t"
                 "s_mov_b64 exec, -1"
                 : "=&v"(seen) : "v"(addr), "v"(px), "s"(edges) : "memory");
    return seen;
}
// The same with ALL 64 lanes: a repeat pixel (px == prev) swaps the value in that the pixel before it - an
// edge pixel, or a repeat pixel for which the same holds - has left in that very slot, so its swap changes
// nothing (no pixel lies between the two) and what it gets back is not looked at.  Two scalar instructions
// less per step.  Not valid where lanes hold no pixel (beyond the image's end) nor before the image's first
// edge (repeats of the start value {0,0,0,255} of qoi.h:396-399, which was never written to the table): the
// GEN form of the step uses probe_swap().
__device__ __forceinline__ uint32_t probe_swap_all(uint32_t addr, uint32_t px) {
    uint32_t seen;
    asm volatile("ds_wrxchg_rtn_b32 %0, %1, %2" : "=&v"(seen) : "v"(addr), "v"(px) : "memory");
    return seen;
}
// The same for callers that do not look at what comes back (the entry-state replays): every exchange returns into the SAME register,
// which stays live from the first exchange to probe_wait() behind the last.  (With the result thrown away the compiler is free to
// hand its register to something else right behind the asm block - it does not know that the instruction writes it LATER, when the
// LDS answers - and whatever then lives there is overwritten: round 5's first state look-back lost half of an address that way.)
__device__ __forceinline__ void probe_swap_into(uint32_t& chain, uint32_t addr, uint32_t px, u64 edges) {
    asm volatile("s_mov_b64 exec, %3\n\t"
                 "ds_wrxchg_rtn_b32 %0, %1, %2\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(chain) : "v"(addr), "v"(px), "s"(edges) : "memory");
}
__device__ __forceinline__ void probe_wait(uint32_t& seen) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(seen) : : "memory"); }

// w = (seen == px ? idx : we) in the lanes of `edges`, unchanged elsewhere.  Waits for the probe's LDS result first.
// (The scalar instructions between the compare and the select are also the two wait states gfx950 wants between a VALU
// write of vcc and a VALU read of it.)
__device__ __forceinline__ void select_edge_word(uint32_t& w, uint32_t we, uint32_t idx, uint32_t seen, uint32_t px, u64 edges) {
    asm volatile("s_waitcnt lgkmcnt(0)\n\t"
                 "v_cmp_eq_u32 vcc, %3, %4\n\t"
                 "s_mov_b64 exec, %5\n\t"
                 "s_nop 0\n\t"
                 "v_cndmask_b32 %0, %1, %2, vcc\n\t"
                 "s_mov_b64 exec, -1"
                 : "+v"(w) : "v"(we), "v"(idx), "v"(seen), "v"(px), "s"(edges) : "vcc", "memory");
}
// byte k of a minus byte k of b in the low byte of the result (upper bits: don't care)
__device__ __forceinline__ uint32_t sub_byte1(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_1 src1_sel:BYTE_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ uint32_t sub_byte2(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("v_sub_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_2 src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// repeat lanes (NE) whose predecessor is a repeat as well; lane 0's predecessor is the last pixel of the step before: a repeat
// unless ccp == 64.  Scalar unit only (written in C the compiler moved the whole test to the vector unit: seven instructions).
__device__ __forceinline__ u64 repeats_side_by_side(u64 NE, uint32_t ccp) {
    u64 r, t;
    asm("s_cmp_lg_u32 %2, 64\n\t"
        "s_cselect_b64 %0, 1, 0\n\t"
        "s_lshl_b64 %1, %3, 1\n\t"
        "s_or_b64 %0, %0, %1\n\t"
        "s_and_b64 %0, %0, %3"
        : "=&s"(r), "=&s"(t) : "s"(__builtin_amdgcn_readfirstlane((int)ccp)), "s"(NE) : "scc");   // (ccp is wave-uniform: free where it already sits in an SGPR)
    return r;
}

// chunk word tags (above the bytes a short chunk stores): length class of the lane.  1-byte chunks need no tag: nothing tests
// for them.  A QOI_OP_LUMA word comes out of v_perm_b32 with byte 3 = 0xFF (the only constants it offers are 0x00 and 0xFF):
// two-byte words are the NEGATIVE ones, the long marker is 0x40000000 (the inline constant 2.0).
constexpr uint32_t kLenLong = 0x40000000u;
__device__ __forceinline__ bool word_is_two(uint32_t w) { return (int32_t)w < 0; }     // (long words excluded by the caller where they can occur)
__device__ __forceinline__ bool word_is_long(uint32_t w) { return (int32_t)w >= (int32_t)kLenLong; }

// ---- the literal classes of TWO steps at once ------------------------------------------------------------------------------
// The biased deltas of an even step and of the odd step after it share registers as 16-bit halves (the sign-extending SDWA adds
// that make them anyway write WORD_0 / WORD_1), so the range tests' ORs and shifts, the subtractions for LUMA and the packing of
// the DIFF / LUMA bytes are ONE packed instruction (v_pk_sub_u16, v_pk_lshrrev_b16, v_pk_mad_u16, v_or3) for both steps: 17
// vector instructions per step for "deltas, tests, words" instead of 22.  The halves are taken apart again by the consumers'
// operand selects (v_cmp_*_sdwa WORD_k, v_cndmask_b32_sdwa WORD_k, v_perm_b32), which cost nothing.
struct PairClass {
    uint32_t od2;   // tr | tg | tb per half (t = wrapped delta + 2): QOI_OP_DIFF iff < 4          (qoi.h:446-453)
    uint32_t ol2;   // (ug >> 2) | ur | ub per half: QOI_OP_LUMA iff < 16                             (qoi.h:455-459)
    uint32_t wd2;   // the DIFF byte 0x40 | tr << 4 | tg << 2 | tb per half
    uint32_t b0p;   // the LUMA bytes per half: 0x80 | ug ...
    uint32_t b1p;   // ... and ur << 4 | ub
};
typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ u16x2 as_h2(uint32_t v) { return __builtin_bit_cast(u16x2, v); }
__device__ __forceinline__ uint32_t as_u32(u16x2 v) { return __builtin_bit_cast(uint32_t, v); }
__device__ __forceinline__ uint32_t pk_mad_u16(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_pk_mad_u16 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b), "v"(c));
    return r;
}
__device__ __forceinline__ void classify_pair(PairClass& K, uint32_t px0, uint32_t pv0, uint32_t px1, uint32_t pv1) {
    // wrapped byte deltas in the low byte of d* (upper bits: don't care)
    const uint32_t d0r = px0 - pv0, d0g = sub_byte1(px0, pv0), d0b = sub_byte2(px0, pv0);
    const uint32_t d1r = px1 - pv1, d1g = sub_byte1(px1, pv1), d1b = sub_byte2(px1, pv1);
    uint32_t tr2, tg2, tb2, tg82, ug2;
    // half k = constant + sign-extended delta byte of step k.  (An instruction that writes part of a register must not be
    // followed at once by one that reads the register: the five of the even step stand between every pair; s_nop at the end.)
    asm("v_add_u32_sdwa %0, %5, sext(%8) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_sdwa %1, %5, sext(%9) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_sdwa %2, %5, sext(%10) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_sdwa %3, %6, sext(%9) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_sdwa %4, %7, sext(%9) dst_sel:WORD_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_sdwa %0, %5, sext(%11) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_sdwa %1, %5, sext(%12) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_sdwa %2, %5, sext(%13) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_sdwa %3, %6, sext(%12) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "v_add_u32_sdwa %4, %7, sext(%12) dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:BYTE_0\n\t"
        "s_nop 0"
        : "=&v"(tr2), "=&v"(tg2), "=&v"(tb2), "=&v"(tg82), "=&v"(ug2)
        : "s"(2u), "s"(0xFFFFFFFAu), "s"(32u), "v"(d0r), "v"(d0g), "v"(d0b), "v"(d1r), "v"(d1g), "v"(d1b));
    const uint32_t ur2 = as_u32(as_h2(tr2) - as_h2(tg82)), ub2 = as_u32(as_h2(tb2) - as_h2(tg82));    // dr-dg+8, db-dg+8
    K.od2 = tr2 | tg2 | tb2;
    K.ol2 = as_u32(as_h2(ug2) >> (u16x2)(2)) | ur2 | ub2;
    K.wd2 = pk_mad_u16(tr2, 0x00100010u, pk_mad_u16(tg2, 0x00040004u, tb2)) | 0x00400040u;
    K.b0p = ug2 | 0x00800080u;
    K.b1p = pk_mad_u16(ur2, 0x00100010u, ub2);
}
// The literal chunk word of step HALF of the pair (qoi.h:438-474): QOI_OP_LUMA (byte 0 in bits 0..7, byte 1 in bits 16..23,
// 0xFF on top) where its test holds, QOI_OP_DIFF where that one holds, the long marker where neither does or the alpha moved.
// Also returns the lanes whose alpha moved.  Hand-scheduled: a scalar pair written by a vector compare is read two instructions
// later at the earliest.
template <int HALF>
__device__ __forceinline__ uint32_t literal_word(const PairClass& K, uint32_t px, uint32_t prev, u64& m_ad) {
    uint32_t we;
    u64 s_luma;
    if (HALF == 0) {
        asm("v_cmp_gt_u32_sdwa %1, %3, %5 src0_sel:DWORD src1_sel:WORD_0\n\t"
            "v_cmp_gt_u32_sdwa vcc, %4, %6 src0_sel:DWORD src1_sel:WORD_0\n\t"
            "v_perm_b32 %0, %8, %7, %9\n\t"
            "v_cmp_ne_u32_sdwa %2, %10, %11 src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"
            "v_cndmask_b32 %0, 2.0, %0, %1\n\t"
            "v_cndmask_b32_sdwa %0, %0, %12, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n\t"
            "v_cndmask_b32 %0, %0, 2.0, %2"
            : "=&v"(we), "=&s"(s_luma), "=&s"(m_ad)
            : "s"(16u), "s"(4u), "v"(K.ol2), "v"(K.od2), "v"(K.b0p), "v"(K.b1p), "s"(0x0D040C00u), "v"(px), "v"(prev), "v"(K.wd2) : "vcc");
    } else {
        asm("v_cmp_gt_u32_sdwa %1, %3, %5 src0_sel:DWORD src1_sel:WORD_1\n\t"
            "v_cmp_gt_u32_sdwa vcc, %4, %6 src0_sel:DWORD src1_sel:WORD_1\n\t"
            "v_perm_b32 %0, %8, %7, %9\n\t"
            "v_cmp_ne_u32_sdwa %2, %10, %11 src0_sel:BYTE_3 src1_sel:BYTE_3\n\t"
            "v_cndmask_b32 %0, 2.0, %0, %1\n\t"
            "v_cndmask_b32_sdwa %0, %0, %12, vcc dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1\n\t"
            "v_cndmask_b32 %0, %0, 2.0, %2"
            : "=&v"(we), "=&s"(s_luma), "=&s"(m_ad)
            : "s"(16u), "s"(4u), "v"(K.ol2), "v"(K.od2), "v"(K.b0p), "v"(K.b1p), "s"(0x0D060C02u), "v"(px), "v"(prev), "v"(K.wd2) : "vcc");
    }
    return we;
}

// per-lane constants of the step
struct LaneConst {
    uint32_t below_lo, below_hi;   // masks of the lanes below this one
    uint32_t lane_run;             // lane + 128: + clz(edges below) = 0xBF + run length
    uint32_t tbase;                // LDS address of the colour table (256-byte aligned), kept in a VGPR: slot address = one v_and_or
};

// ---- one step: the chunks of 64 pixels -----------------------------------------------------------
// px / prev: this lane's pixel and the one before it.  Ec: edge lanes (px != prev) of this step.  nb63: bit 63 set if the
// pixel after lane 63 is an edge.  GEN only: V valid lanes, lastbit the lane of the image's last pixel.
// ccp (scalar) = 63 + (first pixel of the step - last edge before the step): stands in for clz(edges below the lane).
// vbase (same value in every lane): LDS address of the next staged byte.
template <int PROBE, bool GEN, int HALF, class LDS>
__device__ __forceinline__ void encode_step(LDS& L, const LaneConst& C, uint32_t lane, uint32_t px, uint32_t prev, const PairClass& K,
                                            u64 Ec, u64 nb63, u64 V, u64 lastbit, uint32_t& ccp, uint32_t& vbase) {
        const u64 En = (Ec >> 1) | nb63 | lastbit;             // lanes whose successor is an edge (or that end the image)
    const u64 NE = GEN ? (~Ec & V) : ~Ec;                  // repeat pixels
    u64 RB = NE & En;                                      // repeat pixels that close a run: they carry its run byte

    // ---- repeats: run byte 0xC0|(run-1) on the pixel that closes a run (qoi.h:416-421,425-428) ----
    // clz of the edges below the lane; ccp stands in when the run began before this step
    // A repeat pixel whose predecessor is an edge counts 1: where no two repeats stand side by side (and the step does not begin
    // inside a run: ccp == 64 says the pixel before it was an edge) every run byte of the step is 0xC0 and the eight vector
    // instructions of the count are skipped on a scalar test - three steps in four of a photograph.
    uint32_t w = 0xC0u;
#ifndef QOIMI_ENC_NO_SHORT_RUNS
    if (repeats_side_by_side(NE, ccp))
#endif
    {
        const uint32_t fhi = ffbh((uint32_t)(Ec >> 32) & C.below_hi);
        const uint32_t flo = ffbh((uint32_t)Ec & C.below_lo) | 32u;
        const uint32_t m = min(min(fhi, flo), ccp);
        w = m + C.lane_run;                                // 0xBF + count, count = repeats since the last edge
        if (__ballot(w > 0xFCu) & NE) {                    // some run reaches 62: wave-uniform slow path (flat content)
            const uint32_t cnt = w - 0xBFu;
            const uint32_t xm = cnt % 62u;
            w = xm ? 0xBFu + xm : 0xFDu;                   // a repeat landing on a multiple of 62 closes a full run
            RB |= NE & __ballot(xm == 0u);
        }
    }
    // (Control flow of the step: plain `if` blocks only, no `else`.  The compiler restructures every if / else into two
    // consecutive ifs on a flag even where the condition is wave-uniform - five scalar instructions per step.)
    u64 any = Ec | RB;                                     // lanes that emit a chunk
    ccp = scalar_add64(ccp);                               // no edge in this step: the last edge is 64 pixels further away
    if (Ec) {
        ccp = clz64_plus64(Ec);
        // ---- colour-table probe/update (qoi.h:430-436) for edge pixels ---------------------
        const uint32_t hsh = __builtin_amdgcn_udot4(px, 0x2C1C140Cu, 0u, false);      // 4 * QOI_COLOR_HASH (qoi.h:322)
        uint32_t seen = ~px;
        if (PROBE == 1) {
            seen = GEN ? probe_swap((hsh & 0xFCu) | C.tbase, px, Ec) : probe_swap_all((hsh & 0xFCu) | C.tbase, px);
        } else {
            const uint32_t so = hsh & 0xFCu;
            const bool edge = in_mask(Ec);
            const u64 lane_bit = 1ull << lane;
            if (edge) __hip_atomic_fetch_or(&L.mask[so >> 2], lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            __builtin_amdgcn_wave_barrier();
            const u64 same = edge ? L.mask[so >> 2] : 0ull;    // edge lanes of this step sharing the slot
            const uint32_t tval = L.table[so >> 2];
            __builtin_amdgcn_wave_barrier();
            if (edge) L.mask[so >> 2] = 0;
            const u64 pred = same & (lane_bit - 1ull);
            const uint32_t pv = gather_lane(px, pred ? (uint32_t)msb64(pred) : lane);
            seen = pred ? pv : tval;                           // nearest earlier same-slot edge, else carried table
            if (edge && ((same >> lane) >> 1) == 0) L.table[so >> 2] = px;   // last lane per slot updates the table
            __builtin_amdgcn_wave_barrier();
        }
        // ---- chunk of an edge pixel (qoi.h:432-474): INDEX, else RGBA if alpha moved, else DIFF, LUMA, RGB ----
        u64 m_ad;                                          // lanes whose alpha differs from the previous pixel's
        const uint32_t we = literal_word<HALF>(K, px, prev, m_ad);
        // QOI_OP_INDEX (qoi.h:432-434) where the slot held the pixel; the edge lanes take their chunk word, the others keep
        // their run byte (one v_cndmask under exec = edges instead of two)
        select_edge_word(w, we, (hsh >> 2) & 63u, seen, px, Ec);
        const u64 lng = __ballot(word_is_long(w));
        if (__builtin_expect(lng != 0ull, 0)) {
            // rare in natural images: some lane carries QOI_OP_RGB / QOI_OP_RGBA (qoi.h:461-474): tag r g b (a)
            const u64 five = lng & m_ad;
            const u64 two = __ballot(word_is_two(w)) & ~lng;
            const u64 b0 = (any & ~(two | lng)) | five;            // odd lengths: 1-byte chunks and RGBA
            const uint32_t off = vbase + count_below(b0) + 2u * count_below(two) + 4u * count_below(lng);
            const bool is_long = in_mask(lng);
            const uint32_t w0 = is_long ? (in_mask(five) ? kTagRgba : kTagRgb) | (px << 16) : w;   // byte 0: tag, bits 16..23: r
            stage_short(off, w0, any, two | lng);
            stage_long_rest(off, px, lng, five);
            vbase += (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(two) + 4u * (uint32_t)__builtin_popcountll(lng);
            any = 0ull;                                            // nothing left for the tail below
        }
    }
    // ---- common case: chunk lengths 1 and 2 only.  offset = #chunks below + #LUMA chunks below ----
    // (skipped where the long-chunk body above has staged everything - content dense in QOI_OP_RGB / QOI_OP_RGBA takes that body in
    // nearly every step: photo_hard 17.45 -> 16.11 ms per 1024 frames, noise 8.04 -> 7.00 per 256, photographs 12.26 -> 12.19,
    // profiles/r05_s12_enc_skiptail.txt)
    if (any != 0ull)
    {
        const u64 two = __ballot(word_is_two(w)) & any;
        const uint32_t off = count_below_from(two, count_below_from(any, vbase));
        stage_short(off, w, any, two);
        vbase += (uint32_t)__builtin_popcountll(any) + (uint32_t)__builtin_popcountll(two);
    }
}

// lanes 0..r-1 (r <= 0: none, r >= 64: all)
__device__ __forceinline__ u64 lanes_upto(int r) { return r >= 64 ? ~0ull : (r <= 0 ? 0ull : (1ull << r) - 1ull); }

// The pair stays RAW in its two registers - (px, pv) for 4 channels, the two loaded dwords (hi, lo) for 3 - and is taken apart by
// unpack_pair() where it is USED (process_group, warm_entry_state): written the other way round, the three instructions that take a
// 3-channel pair apart stood right behind their load in the source, and whether the compiler sank them to the use or kept them
// there - every load of a group followed by its own s_waitcnt vmcnt(0) - hung on unrelated code (round 4: the scratch pool's
// branch in the group loop made a 3-channel batch 31 % slower; `s_waitcnt vmcnt` census of the kernel, EXPERIMENTS.md).
template <int CH>
__device__ __forceinline__ void load_pair_at(const uint8_t* __restrict__ q, int k, uint32_t& a, uint32_t& b) {
    if constexpr (CH == 4) {
        a = reinterpret_cast<const uint32_t*>(q)[k];
        b = reinterpret_cast<const uint32_t*>(q)[k - 1];
    } else {
        struct __attribute__((packed, aligned(1))) U2 { uint32_t x, y; };
        const U2 v = *reinterpret_cast<const U2*>(q + k * 3 - 3);
        a = v.y; b = v.x;
    }
}
template <int CH>
__device__ __forceinline__ void unpack_pair(uint32_t a, uint32_t b, uint32_t& px, uint32_t& pv) {
    if constexpr (CH == 4) { px = a; pv = b; }
    else { pv = b | 0xFF000000u; px = __builtin_amdgcn_alignbit(a, b, 24) | 0xFF000000u; }
}
// the raw pair that unpacks to (px, pv) - for the two single pixels around the end of a set, which are loaded one by one
template <int CH>
__device__ __forceinline__ void pack_pair(uint32_t px, uint32_t pv, uint32_t& a, uint32_t& b) {
    if constexpr (CH == 4) { a = px; b = pv; }
    else { b = (pv & 0x00FFFFFFu) | (px << 24); a = px >> 8; }
}

// ---- a group of kGroupSteps steps whose pixels (px) and previous pixels (pv) sit in registers ----------
// E: edges of the group's first step on entry, of the first step AFTER the group on exit (from nx_px / nx_pv: the first
// step of the next group, or the two pixels around the end of the set).  GEN: rem = pixels of the image left at the
// group's first pixel.
// RAW: 0 = px / pv hold pixels; 3 or 4 = they hold raw pairs of that many channels as load_group leaves them (taken apart here).
template <int PROBE, bool GEN, int RAW, class LDS>
__device__ __forceinline__ void process_group(LDS& L, const LaneConst& C, uint32_t lane,
                                              const uint32_t (&rx)[kGroupSteps], const uint32_t (&rv)[kGroupSteps],
                                              uint32_t nx_rx, uint32_t nx_rv, int rem, u64& E, uint32_t& ccp, uint32_t& vbase) {
    // (a step's pair is taken apart one step ahead of its use - the edges of step t + 1 are wanted in step t - and not before:
    // eight unpacked pairs beside the eight raw ones cost the generic 3-channel instantiation a register spill)
    auto pair_at = [&](int t, uint32_t& px, uint32_t& pv) {
        const uint32_t a = t < kGroupSteps ? rx[t < kGroupSteps ? t : 0] : nx_rx, b = t < kGroupSteps ? rv[t < kGroupSteps ? t : 0] : nx_rv;
        if constexpr (RAW == 3) unpack_pair<3>(a, b, px, pv); else { px = a; pv = b; }
    };
    if (GEN) E &= lanes_upto(rem);                         // (the group before this one does not know where the image ends)
    PairClass K = {0u, 0u, 0u, 0u, 0u};
    uint32_t cpx, cpv, npx, npv;
    pair_at(0, cpx, cpv);
#pragma unroll
    for (int t = 0; t < kGroupSteps; ++t) {
        const u64 Ec = E;
        u64 V = ~0ull, lastbit = 0ull;
        if (GEN) {
            const int r = rem - t * 64;                    // pixels of the image left at this step
            V = lanes_upto(r);
            lastbit = (r >= 1 && r <= 64) ? 1ull << (r - 1) : 0ull;
        }
        // edges of the next step (its lane 0 tells lane 63 whether its run ends here)
        pair_at(t + 1, npx, npv);
        E = __ballot(npx != npv);
        if (GEN) E &= lanes_upto(rem - (t + 1) * 64);
        const u64 nb63 = E << 63;
        if ((t & 1) == 0 && (Ec | E) != 0ull) classify_pair(K, cpx, cpv, npx, npv);   // this step and the next one
        if (!(GEN && V == 0ull)) {
            if (t & 1) encode_step<PROBE, GEN, 1>(L, C, lane, cpx, cpv, K, Ec, nb63, V, lastbit, ccp, vbase);
            else encode_step<PROBE, GEN, 0>(L, C, lane, cpx, cpv, K, Ec, nb63, V, lastbit, ccp, vbase);
        }
        cpx = npx; cpv = npv;
    }
}

// pixels base + 64 t + lane and the pixels before them, t = 0..kGroupSteps-1, of a group that lies inside the image and does not
// hold its first pixel, as RAW pairs: no bounds checks, one 64-bit address, the loads differ in their immediate offsets only
template <int CH>
__device__ __forceinline__ void load_group(const uint8_t* __restrict__ pix, uint32_t base, uint32_t lane,
                                           uint32_t (&px)[kGroupSteps], uint32_t (&pv)[kGroupSteps]) {
    const uint8_t* __restrict__ q = pix + (size_t)(base + lane) * (size_t)CH;
#pragma unroll
    for (int t = 0; t < kGroupSteps; ++t) load_pair_at<CH>(q, t * 64, px[t], pv[t]);
}
// pixel i and the one before it, any i: lanes beyond the image's last pixel get 0, the pixel before the image's first one is
// the start value of qoi.h:396-399
template <int CH>
__device__ __forceinline__ void load_pair_guarded(const uint8_t* __restrict__ pix, uint32_t i, uint32_t n, uint32_t& px, uint32_t& pv) {
    px = i < n ? load_px<CH>(pix, i) : 0u;
    pv = (i < n && i > 0u) ? load_px<CH>(pix, i - 1u) : kInitPx;
}

// Everything a set reads from global memory before its first step besides its first group of pixels.
struct SetIn {
    uint32_t warm[8];            // ENTRY 1: the 512 pixels before the set (step k: pixels lo-64(k+1) .. +63) ...
    uint32_t warm_prev[8];       // ... and the pixel before each of them
    uint32_t tab_loc, tab_far;   // ENTRY 0: entry colour table: group-local part / image-level part (lane = slot)
    u64 tab_valid;
    int le_loc, le_far;          // ENTRY 0: last edge before the set: group-local / image-level
};

// PIPE (round 5 experiment, env QOIMI_ENC_PIPE=1 with QOIMI_ENC_PERSIST): a wavefront that encodes set after set asks for the NEXT set's
// first loads - its look-back window and its first group of pixels, and the ticket that names it - when its current set's groups are
// through, in front of that set's placement: they travel while the wavefront waits for its place and copies its bytes out.  (With
// start-order tickets a wavefront may hold two sets at once: whatever a set waits for was taken EARLIER than either of its holder's two.)
struct SetPre {
    uint32_t warm[8], warm_prev[8];
    uint32_t img, set;
    bool have_ticket;      // (img, set) is this wavefront's next set: its ticket is taken
    bool valid;            // ... and its loads are on their way in the arrays above
};

// ENTRY 1: a set finds its entry state itself.  The colour table before pixel `lo` is "the last edge pixel
// per hash slot" (qoi.h:430-436), so the wavefront walks BACKWARDS over the pixels before its set, 64 at a
// time, and fills every slot that is still empty with the latest edge pixel that hashes there, until all 64
// slots are known or the image start is reached (untouched slots are then the zeroes of qoi.h:393).  Natural
// images and noise need 5-8 steps (SURVEY: all 64 slots are rewritten within ~700 pixels); flat content does
// not finish within the window - the set then flags its image and the generic passes (per-slab summaries +
// scans, ENTRY 0) redo that image.  Unfilled slots hold slot+1, a value that cannot hash to its own slot
// (3(s+1) != s mod 64).  Returns false if the window did not suffice.
constexpr int kWarmSteps = 128;      // look-back window: 8192 pixels (natural content is done after 5-11 steps)
constexpr int kWarmBatch = 8;        // 64-pixel steps loaded together
constexpr int kWarmMinFilled = 48;   // slots that must be known after the first batch (512 pixels), else the content is flat: give up

template <int CH, int PROBE, class LDS>
__device__ __forceinline__ bool warm_entry_state(const uint8_t* __restrict__ pix, uint32_t lo, uint32_t lane,
                                                 LDS& L, uint32_t tbase, const SetIn& in, int& last_edge) {
    last_edge = -1;
    if (lo == 0u) { L.table[lane] = 0u; return true; }       // qoi.h:393: zeroed table, no edge yet
    const uint32_t sent = lane + 1u;
    uint32_t chain = 0u;                                      // what the exchanges return (probe_swap_into)
    L.table[lane] = sent;
    __builtin_amdgcn_wave_barrier();
    // ---- the 512 pixels right before the set, oldest first: later edge pixels simply overwrite earlier ones
    //      (these loads were issued ahead of the set's own pixels) ------------------------------------------
#pragma unroll
    for (int k = kWarmBatch - 1; k >= 0; --k) {
        const int base = (int)lo - 64 * (k + 1);
        uint32_t px, pvw;
        unpack_pair<CH>(in.warm[k], in.warm_prev[k], px, pvw);
        const u64 E = __ballot(px != pvw);
        if (E) {
            last_edge = base + msb64(E);
            probe_swap_into(chain, (__builtin_amdgcn_udot4(px, 0x2C1C140Cu, 0u, false) & 0xFCu) | tbase, px, E);
        }
    }
    probe_wait(chain);
    __builtin_amdgcn_wave_barrier();
    u64 filled = __ballot(L.table[lane] != sent);
    bool full = filled == ~0ull, at_start = (int)lo - 64 * kWarmBatch <= 0;
    if (!full && !at_start && __builtin_popcountll(filled) < kWarmMinFilled) return false;   // flat content: generic path
    // ---- rarely: further back, 64 pixels at a time, filling only slots that are still empty -------------------
#pragma unroll 1
    for (int t0 = kWarmBatch + 1; t0 <= kWarmSteps && !full && !at_start; t0 += kWarmBatch) {
        uint32_t wp[kWarmBatch];
#pragma unroll
        for (int k = 0; k < kWarmBatch; ++k) {
            const int i = (int)lo - 64 * (t0 + k) + (int)lane;
            wp[k] = i >= 0 ? load_px<CH>(pix, (uint32_t)i) : kInitPx;
        }
        const int ci = (int)lo - 64 * (t0 + kWarmBatch - 1) - 1;
        const uint32_t carry = __builtin_amdgcn_readfirstlane(ci >= 0 ? load_px<CH>(pix, (uint32_t)ci) : kInitPx);
#pragma unroll
        for (int k = 0; k < kWarmBatch; ++k) {
            if (full || at_start) break;
            const int base = (int)lo - 64 * (t0 + k);
            const uint32_t px = wp[k];
            const uint32_t prev = k + 1 < kWarmBatch ? prev_pixels(px, wp[k + 1 < kWarmBatch ? k + 1 : k]) : from_lane_below(px, carry);
            const bool edge = px != prev;
            const u64 E = __ballot(edge);
            if (last_edge < 0 && E) last_edge = base + msb64(E);
            const uint32_t so = slot_byte_offset(px);
            const uint32_t cur = *(const lds_u32*)(tbase | so);
            const u64 want = __ballot(edge && cur == (so >> 2) + 1u);
            if (want) { probe_swap_into(chain, tbase | so, px, want); probe_wait(chain); }      // same slot twice in a step: the later pixel wins
            __builtin_amdgcn_wave_barrier();
            filled = __ballot(L.table[lane] != sent);
            full = filled == ~0ull;
            at_start = base <= 0;
        }
    }
    if (at_start && !full) {                                  // image start reached: what is left is the zeroed table
        const uint32_t v = L.table[lane];
        if (v == sent) L.table[lane] = 0u;
        full = true;
    }
    return full && (last_edge >= 0 || at_start);
}

// n bytes from a 16-byte aligned source in global memory to dst (any alignment): head up to the first 16-byte boundary byte by
// byte, aligned 16-byte stores (source re-aligned with v_alignbyte), tail byte by byte.  One wavefront.
// The source is read with L1-bypassing (nt) loads: scratch slots of the pool are REUSED inside one launch, and this CU's vector L1
// may still hold the lines of a slot as an earlier holder on this CU read them - a wavefront's own stores go through to the L2 and
// do not refresh them (MI355X_MICROARCH.md, inter-workgroup visibility).  With a slot per set nothing was ever read twice.
__device__ __forceinline__ void copy_global_out(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, uint32_t n, uint32_t lane) {
    const uint32_t mis = (uint32_t)(uintptr_t)dst & 15u;
    const uint32_t head = min(n, (16u - mis) & 15u);                 // bytes before dst becomes 16-byte aligned
    if (lane < head) dst[lane] = __builtin_nontemporal_load(&src[lane]);
    const uint32_t n16 = (n - head) >> 4;
    uint4* __restrict__ d16 = reinterpret_cast<uint4*>(dst + head);
    const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(src) + (head >> 2);
    const uint32_t sh = head & 3u;
    for (uint32_t j = lane; j < n16; j += 64u) {
        const uint32_t* q = s32 + 4u * j;
        const uint32_t w0 = __builtin_nontemporal_load(&q[0]), w1 = __builtin_nontemporal_load(&q[1]), w2 = __builtin_nontemporal_load(&q[2]),
                       w3 = __builtin_nontemporal_load(&q[3]), w4 = __builtin_nontemporal_load(&q[4]);
        uint4 v;
        v.x = __builtin_amdgcn_alignbyte(w1, w0, sh); v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
        v.z = __builtin_amdgcn_alignbyte(w3, w2, sh); v.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
        d16[j] = v;
    }
    const uint32_t done = head + (n16 << 4);
    if (lane < n - done) dst[done + lane] = __builtin_nontemporal_load(&src[done + lane]);
}
// the same from the wavefront's LDS staging buffer
__device__ __forceinline__ void copy_stage_out(const uint32_t* stage, uint8_t* __restrict__ dst, uint32_t n, uint32_t lane) {
    const uint8_t* stage8 = reinterpret_cast<const uint8_t*>(stage);
    const uint32_t mis = (uint32_t)(uintptr_t)dst & 15u;
    const uint32_t head = min(n, (16u - mis) & 15u);
    if (lane < head) dst[lane] = stage8[lane];
    const uint32_t n16 = (n - head) >> 4;
    uint4* d16 = reinterpret_cast<uint4*>(dst + head);
    const uint32_t sh = head & 3u;
    for (uint32_t j = lane; j < n16; j += 64u) {
        const uint32_t* q = &stage[(head >> 2) + 4u * j];
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
        uint4 v;
        v.x = __builtin_amdgcn_alignbyte(w1, w0, sh); v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
        v.z = __builtin_amdgcn_alignbyte(w3, w2, sh); v.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
        d16[j] = v;
    }
    const uint32_t done_b = head + (n16 << 4);
    if (lane < n - done_b) dst[done_b + lane] = stage8[done_b + lane];
}

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
constexpr int kAuxWriteThrough = 1 | 16;                       // cache policy bits of a buffer instruction on gfx950: sc0 | sc1
// Moves the staged bytes [0, spos) to the set's scratch slot behind the `spilled` bytes already there (a multiple of 16).
// all = false: whole 16-byte pieces only, the remainder moves to the front of the staging buffer.  Returns the bytes left staged.
// The stores are WRITE-THROUGH (sc0 sc1): a pool slot is written by one holder after another, from different XCDs, and a plain store
// leaves a dirty line in its XCD's L2 for as long as that L2 likes - an earlier holder's line written back after the next holder's
// bytes have reached memory would put stale bytes under that holder's read-back (the L2s of the XCDs are not coherent with each
// other, MI355X_MICROARCH.md).  Written through, a holder's bytes are in memory when its `s_waitcnt vmcnt(0)` in front of the
// copy-out returns, i.e. before it gives the slot back; 16-byte sc1 stores cost what plain ones do.
template <int PROBE, class LDS>
__device__ __forceinline__ uint32_t spill_stage(LDS& L, uint8_t* __restrict__ slot, uint32_t& spilled, uint32_t spos, bool all, uint32_t lane) {
    const uint32_t n16 = all ? (spos + 15u) >> 4 : spos >> 4;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)slot, 0, 0x7FFFFFF0, 0x00020000);
    const u32x4* src = reinterpret_cast<const u32x4*>(L.stage);
    __builtin_amdgcn_wave_barrier();
    for (uint32_t j = lane; j < n16; j += 64u) __builtin_amdgcn_raw_buffer_store_b128(src[j], rs, spilled + (j << 4), 0, kAuxWriteThrough);
    spilled += n16 << 4;
    if (all) return 0u;
    const uint32_t keep = lane < 4u ? L.stage[(n16 << 2) + lane] : 0u;     // the incomplete piece (LDS ops of a wavefront run in order)
    __builtin_amdgcn_wave_barrier();
    if (lane < 4u) L.stage[lane] = keep;
    __builtin_amdgcn_wave_barrier();
    return spos & 15u;
}

// Scratch slots of the sets that spill (look-back mode).  A worst-case slot per set was 42.5 GB for the 1024-frame 4K shard and stayed
// untouched on photographs; only the sets in flight ever hold spilled bytes.  A set takes a slot at its first spill (a bit of the
// map, lane 0) and gives it back after its copy-out - by then its loads from the slot have all returned (their data went into the
// stores).  All-zero = all free: the map is part of the header the launcher zeroes.
__device__ __forceinline__ uint32_t pool_take(const EncParams& p, uint32_t seed, uint32_t lane) {
    uint32_t id = p.pool_slots;                                // the emergency slot: only reached with every slot taken (never expected)
    if (lane == 0) {
        // One word of the map per 128-byte line (with the 128 words of 8192 slots in eight lines, every set of a noise batch queued at
        // the same few lines for its slot: 36 us per set).  The first fetch-or is a guess; what it returns names the word's free bits.
        const uint32_t words = p.pool_slots >> 6;
        uint32_t w = seed % words, bit = (seed >> 16) & 63u;
        for (uint32_t tries = 0; tries < 64u * words; ++tries) {
            const u64 old = __hip_atomic_fetch_or((gu64*)&p.pool_map[(size_t)w * kEncPoolMapStride], 1ull << bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (((old >> bit) & 1ull) == 0ull) { id = w * 64u + bit; break; }
            const u64 free_bits = ~(old | (1ull << bit));
            if (free_bits == 0ull) { w = w + 1u < words ? w + 1u : 0u; continue; }         // (the bit set above was set before: nothing to undo)
            const uint32_t r = (bit + 17u) & 63u;                                         // the next free bit from a place of our own
            const u64 rot = (free_bits >> r) | (r ? free_bits << (64u - r) : 0ull);
            bit = ((uint32_t)__builtin_ctzll(rot) + r) & 63u;
        }
        if (id == p.pool_slots) atomicOr(p.err, 2u);
    }
    return (uint32_t)__builtin_amdgcn_readfirstlane((int)id);
}
__device__ __forceinline__ void pool_give(const EncParams& p, uint32_t id, uint32_t lane) {
    if (lane == 0 && id < p.pool_slots)
        (void)__hip_atomic_fetch_and((gu64*)&p.pool_map[(size_t)(id >> 6) * kEncPoolMapStride], ~(1ull << (id & 63u)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ENTRY 2 (round 5): the entry state of a set of a FLAGGED image (flat content: the look-back window of ENTRY 1 does not determine
// the colour table) by decoupled look-back over the SETS of the image - no summary pass over the pixels, no scans, one launch:
//   * the set first walks its own pixels for "the last edge pixel per hash slot" and its last edge (what enc_slab_summary computed
//     for every slab in a pass of its own: every pixel read from HBM twice) and publishes that as 65 eight-byte granules
//     (slot k: word | valid << 32, granule 64: last edge + 1), state LOCAL;
//   * then it looks back over the sets in front of it, eight at a time: a slot takes the nearest valid LOCAL word, or whatever an
//     INCLUSIVE granule says (the table as it stands behind that set); the last edge likewise.  Every granule carries its state and
//     the call's epoch ("the data is the flag", qoi_dev.h): nothing to zero between calls, no fences, a reader may see a set's
//     granules half LOCAL, half INCLUSIVE - each slot resolves on its own;
//   * it publishes its own INCLUSIVE granules and encodes - the second walk over its pixels finds them in the L2 / Infinity Cache.
// Sets are handed out in start order (tickets) or by workgroup index (tree calls) exactly as the placement look-back wants them.
constexpr u64 kG2Local = 1ull << 62, kG2Incl = 2ull << 62;
__device__ __forceinline__ u64 g2_tag(const EncParams& p) { return (u64)(p.epoch & 0x1FFFFFFFu) << 33; }
__device__ __forceinline__ bool g2_ready(const EncParams& p, u64 g) { return ((g >> 33) & 0x1FFFFFFFull) == (u64)(p.epoch & 0x1FFFFFFFu) && (g >> 62) != 0ull; }

template <int CH, class LDS>
__device__ __forceinline__ bool g2_entry_state(const EncParams& p, const uint8_t* __restrict__ pix, uint32_t n, uint32_t lo, uint32_t hi,
                                               u64* __restrict__ rec_img, uint32_t set, uint32_t lane, LDS& L, uint32_t tbase, int& last_edge, bool& no_edges,
                                               bool* front_walked = nullptr) {
    // ---- own summary: last edge pixel per slot, last edge -------------------------------------------------------------
    const uint32_t sent = lane + 1u;                           // cannot hash to its own slot (warm_entry_state)
    L.table[lane] = sent;
    __builtin_amdgcn_wave_barrier();
    int le_loc = -1;
    const uint32_t ngroups = (hi - lo + kGroupPx - 1u) / kGroupPx;
    // One load per pixel: the pixel in front of a lane's is its neighbour's (a DPP move; lane 0 takes lane 63 of the step before).  (The
    // first form asked for every pixel AND the one before it, bounds-checked, as the encoding groups do: 34 GB of constant frames in
    // 12.2 ms = 2.8 TB/s, profiles/r05_s6_enc_state_lookback_tune.txt.)  3-channel pixels as one unaligned dword where a byte
    // behind the pixel still belongs to the image.
    auto load_one = [&](uint32_t i) -> uint32_t {
        if constexpr (CH == 4) return load_px<4>(pix, i);
        else {
            if (i + 1u < n) { struct __attribute__((packed, aligned(1))) U1 { uint32_t x; }; return (reinterpret_cast<const U1*>(pix + (size_t)i * 3u)->x & 0x00FFFFFFu) | 0xFF000000u; }
            return load_px<3>(pix, i);
        }
    };
    uint32_t cx[kGroupSteps], nx[kGroupSteps];
    auto fetch = [&](uint32_t g, uint32_t (&x)[kGroupSteps]) {
        const uint32_t base = lo + g * kGroupPx;
#pragma unroll
        for (int t = 0; t < kGroupSteps; ++t) { const uint32_t i = base + (uint32_t)t * 64u + lane; x[t] = i < hi ? load_one(i) : 0u; }
    };
    uint32_t chain = 0u;
    // a group of the set into the table at `tb`: the last edge pixel per slot, the last edge
    auto walk = [&](uint32_t g, const uint32_t (&x)[kGroupSteps], uint32_t tb, int& le, uint32_t& carry) {
        const uint32_t base = lo + g * kGroupPx;
        // (what the exchanges return is of no interest - but see probe_swap_into)
#pragma unroll
        for (int t = 0; t < kGroupSteps; ++t) {
            const bool inb = base + (uint32_t)t * 64u + lane < hi;
            const uint32_t prev = from_lane_below(x[t], carry);
            const u64 E = __ballot(inb && x[t] != prev);
            carry = read_lane(x[t], 63);
            if (E) {
                le = (int)(base + (uint32_t)t * 64u) + msb64(E);
                probe_swap_into(chain, tb | slot_byte_offset(x[t]), x[t], E);  // lanes in ascending order: the later pixel stays (PROBE 1)
            }
        }
    };
    // The set's LAST groups first: where they write all 64 slots and hold an edge (photographs, the inside of a sprite) they say everything
    // about the table behind the set, and the groups in front of them are not walked at all (the state look-back read every flagged image
    // twice - 512 soft-alpha sprites: 4 of its 9.7 ms).  Where they do not (flat stretches), the groups in front go into a table of their
    // own - the staging buffer, idle until the encoding starts, 256 bytes behind the table and aligned like it - and fill the gaps; their
    // first group is asked for before the tail has been looked at (512 pixels asked for in vain where the tail does: the walk of the groups
    // in front starts without a load in its way - as a second pipeline of its own it cost 1024 constant frames 7.2 -> 7.8 ms,
    // profiles/r05_s20_enc_tail_first.txt).
    static_assert(kG2TailGroups == 2u, "the tail is the two register groups");
    const uint32_t g_tail = ngroups > kG2TailGroups ? ngroups - kG2TailGroups : 0u;
    const uint32_t tail_first = lo + g_tail * kGroupPx;
    uint32_t carry_tail = tail_first > 0u ? load_one(tail_first - 1u) : kInitPx;     // the pixel in front (qoi.h:396-399 in front of the image)
    uint32_t carry_head = lo > 0u ? load_one(lo - 1u) : kInitPx;
    fetch(g_tail, cx);
    if (g_tail + 1u < ngroups) fetch(g_tail + 1u, nx);
    walk(g_tail, cx, tbase, le_loc, carry_tail);
    if (g_tail != 0u) fetch(0u, cx);
    if (g_tail + 1u < ngroups) walk(g_tail + 1u, nx, tbase, le_loc, carry_tail);
    probe_wait(chain);
    __builtin_amdgcn_wave_barrier();
    uint32_t loc_w = L.table[lane];
    bool loc_valid = loc_w != sent;
    const bool tail_does = lanes_where(loc_valid) == ~0ull && le_loc >= 0;
    if (front_walked) *front_walked = !tail_does;              // (a set of flat stretches: what the first pass would have flagged its image for)
    if (g_tail != 0u && !tail_does) {
        L.stage[lane] = sent;
        __builtin_amdgcn_wave_barrier();
        int le_head = -1;
        for (uint32_t g = 0; g < g_tail; g += 2u) {
            if (g + 1u < g_tail) fetch(g + 1u, nx);
            walk(g, cx, tbase + 256u, le_head, carry_head);
            if (g + 1u >= g_tail) break;
            if (g + 2u < g_tail) fetch(g + 2u, cx);
            walk(g + 1u, nx, tbase + 256u, le_head, carry_head);
        }
        probe_wait(chain);
        __builtin_amdgcn_wave_barrier();
        const uint32_t head_w = L.stage[lane];
        if (!loc_valid) { loc_w = head_w; loc_valid = head_w != sent; }
        if (le_loc < 0) le_loc = le_head;
    }
    u64* const mine = rec_img + (size_t)set * 65u;
    const u64 tag = g2_tag(p);
    // (a set that wrote ALL 64 slots itself - most sets of photograph-like content - leaves a table that owes nothing to the sets in front
    // of it: its first publication is INCLUSIVE already, and the look-backs of the sets behind it end there)
    const bool self_incl = lanes_where(loc_valid) == ~0ull && le_loc >= 0;
    const u64 first_state = self_incl ? kG2Incl : kG2Local;
    granule_store(&mine[lane], (u64)(loc_valid ? loc_w : 0u) | ((u64)(loc_valid ? 1u : 0u) << 32) | tag | first_state);
    if (lane == 0) granule_store(&mine[64], (u64)(uint32_t)(le_loc + 1) | tag | first_state);
    // ---- the sets in front: nearest valid word per slot, nearest edge --------------------------------------------------
    uint32_t ent_w = 0u; bool ent_valid = false, slot_done = loc_valid;      // (a slot this set wrote needs no entry word for the inclusive table - but the ENCODE does: see below)
    int le_ent = -1; bool le_done = false;
    // the encoder needs the entry word of EVERY slot (an edge pixel is compared with what its slot held BEFORE the set), so all 64 are looked up
    slot_done = false;
    uint32_t spins = 0;
    constexpr int kWin = 8;                                    // sets looked at per poll (one frame alone: every set of the image is in flight, the walk back is long)
    for (int j0 = (int)set - 1; j0 >= 0 && (lanes_where(!slot_done) != 0ull || !le_done); j0 -= kWin) {
        for (;;) {
            u64 gw[kWin], gl[kWin];
#pragma unroll
            for (int k = 0; k < kWin; ++k) {
                const int j = j0 - k;
                gw[k] = j >= 0 ? granule_load(&rec_img[(size_t)j * 65u + lane]) : (tag | kG2Incl);       // in front of set 0: the zeroed table, no edge (qoi.h:393)
                gl[k] = j >= 0 ? granule_load(&rec_img[(size_t)j * 65u + 64u]) : (tag | kG2Incl);
            }
            bool all = true;
#pragma unroll
            for (int k = 0; k < kWin; ++k) all = all && g2_ready(p, gw[k]) && g2_ready(p, gl[k]);
            if (lanes_where(!all) == 0ull) {
#pragma unroll
                for (int k = 0; k < kWin; ++k) {
                    const bool incl = (gw[k] >> 62) == 2ull, v = ((gw[k] >> 32) & 1ull) != 0ull;
                    if (!slot_done && (incl || v)) { ent_w = v ? (uint32_t)gw[k] : 0u; ent_valid = v; slot_done = true; }
                    const bool incl_l = (gl[k] >> 62) == 2ull; const int lv = (int)(uint32_t)gl[k] - 1;
                    if (!le_done && (incl_l || lv >= 0)) { le_ent = lv; le_done = true; }
                }
                break;
            }
            if (++spins > p.spin_bound || ((spins & 63u) == 63u && __hip_atomic_load((gu32*)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (lane == 0) atomicOr(p.err, 1u);
                return false;
            }
            __builtin_amdgcn_s_sleep(2);
        }
    }
    le_ent = __builtin_amdgcn_readfirstlane(le_ent);
    // ---- the table as it stands behind this set, for the sets that follow ----------------------------------------------
    {
        const bool v = loc_valid || ent_valid;
        const int le_inc = le_loc >= 0 ? le_loc : le_ent;
        granule_store(&mine[lane], (u64)(loc_valid ? loc_w : (ent_valid ? ent_w : 0u)) | ((u64)(v ? 1u : 0u) << 32) | tag | kG2Incl);
        if (lane == 0) granule_store(&mine[64], (u64)(uint32_t)(le_inc + 1) | tag | kG2Incl);
    }
    __builtin_amdgcn_wave_barrier();
    L.table[lane] = ent_valid ? ent_w : 0u;                   // untouched slots are the zeroes of qoi.h:393
    last_edge = le_ent;
    no_edges = le_loc < 0;
    __builtin_amdgcn_wave_barrier();
    return true;
}


// The first set of a call that finds a flat stretch in front of it says so to the HOST: the call's number into a pinned word, read (without
// a wait) when the context's next calls are set up - a hint, no more: whichever pass a call takes, its streams are the same bytes.  The word
// behind it holds the number of the last call whose first set has STARTED: the host may be hundreds of calls ahead of the device, "no flat
// stretch lately" means lately on the device.
__device__ __forceinline__ void leave_hint(const EncParams& p) {
    if (p.host_hint) __hip_atomic_store(p.host_hint, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

template <int CH, int PROBE, int ENTRY, bool MIXED, bool PIPE = false, class LDS>
__device__ __forceinline__ void encode_set(const EncParams& p, uint32_t img, uint32_t set, uint32_t lane, LDS& L,
                                           SetPre* pre = nullptr, bool use_pre = false, bool has_next = false, uint32_t next_img = 0u) {
    const EncImage I = enc_image<MIXED>(p, img);
    const uint8_t* __restrict__ pix = p.pixels + I.pixel_off;
    const uint32_t n = I.npx;
    const uint32_t lo = set * p.set_px;                        // first pixel of the set (a slab boundary)
    const uint32_t hi = min(n, lo + p.set_px);                 // one past its last pixel
    const bool last_set = hi == n;
    const uint32_t ngroups = (hi - lo + kGroupPx - 1u) / kGroupPx;
    const size_t sg = (size_t)I.set_base + set;                // global index of the set
    if ((ENTRY == 1 || ENTRY == 3) && p.host_hint && sg == 0 && lane == 0)      // how far the device has come: leave_hint
        __hip_atomic_store(p.host_hint + 1, p.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#ifdef QOIMI_ENC_PHASES
    unsigned long long t_mark = __builtin_readcyclecounter();
    if (lane == 0) atomicAdd(&g_enc_phase[5], 1ull);
#endif

    // ---- loads: what the entry state needs first, then the first group of pixels ------------------------------
    // The vector issue of a SIMD goes to the highest priority, then to the OLDEST wavefront (MI355X_MICROARCH.md): a set that has
    // just started loses every arbitration and its first loads go out late.  Raised priority up to the end of the entry state - a few
    // dozen instructions - costs the older wavefronts next to nothing (1024 x 4K photographs: 12.56 -> 12.40 ms, profiles/r04_s3_*).
    __builtin_amdgcn_s_setprio(2);
    SetIn in;
    if (PIPE) {
        // The registers a set's first loads land in are the SAME whether this set asks for them here or the set before it did (SetPre):
        // one definition on either path, nothing to copy at the join (a copy of a value that is still on its way waits for it).
        if (!use_pre && lo != 0u) {
            const uint8_t* __restrict__ q = pix + (size_t)(lo + lane) * (size_t)CH;
#pragma unroll
            for (int k = 0; k < 8; ++k) load_pair_at<CH>(q, -64 * (k + 1), pre->warm[k], pre->warm_prev[k]);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) { in.warm[k] = pre->warm[k]; in.warm_prev[k] = pre->warm_prev[k]; }
    } else if (ENTRY == 1 || ENTRY == 3) {
        if (lo != 0u) {                                        // (a set begins on a slab boundary: lo >= 1024, all of these lie inside the image)
            const uint8_t* __restrict__ q = pix + (size_t)(lo + lane) * (size_t)CH;
#pragma unroll
            for (int k = 0; k < 8; ++k) load_pair_at<CH>(q, -64 * (k + 1), in.warm[k], in.warm_prev[k]);
        }
    } else if (ENTRY == 0) {
        const uint32_t s = set * p.set_slabs;                  // first slab of the set: its entry state is the set's
        const size_t g = (size_t)I.slab_base + s;
        const size_t G = (size_t)I.grp_base + (s >> 6);
        in.tab_loc = p.ent_tab[g * 64u + lane];
        in.tab_far = p.gent_tab[G * 64u + lane];
        in.tab_valid = p.ent_valid[g];
        in.le_loc = p.ent_le[g];
        in.le_far = p.gent_le[G];
    }
    // The pipelined loop below handles the groups that lie inside the image (all 64 lanes of every step valid, the probe may
    // run with all lanes); the image's first set and the group that holds the image's last pixel take the general form after it.
    const bool gen_set = lo == 0u;
    uint32_t nint = gen_set ? 0u : (last_set ? ngroups - 1u : ngroups);
    uint32_t ax[kGroupSteps], av[kGroupSteps], bx[kGroupSteps], bv[kGroupSteps];
    // (PIPE asks ahead for the look-back window only - what a set needs FIRST; its first group goes out here and travels during the replay
    // of the window.  With the group asked for ahead as well the kernel needs 89 registers: spills, or five wavefronts per SIMD.)
    if (ENTRY != 2 && nint) load_group<CH>(pix, lo, lane, ax, av);

    // ---- entry state: colour table + distance to the last edge ---------------------------
    LaneConst C;
    C.below_lo = lane < 32u ? (1u << lane) - 1u : 0xFFFFFFFFu;
    C.below_hi = lane < 32u ? 0u : (1u << (lane - 32u)) - 1u;
    C.lane_run = lane + 128u;
    C.tbase = lds_addr(L.table);                               // 256-byte aligned
    asm volatile("" : "+v"(C.tbase));                          // keep in a VGPR
    int last_edge = -1;
    bool run_only = false;                                     // ENTRY 2: the set holds no edge at all - its bytes are run bytes, known without a second walk
    bool fell_back = false;                                    // ENTRY 3: the look-back window did not do, the set took the state look-back
    if (ENTRY == 1) {
        if (!warm_entry_state<CH, PROBE>(pix, lo, lane, L, C.tbase, in, last_edge)) {
            // (any_generic counts the flagged images: the host reads it behind the call - a batch of flagged images only tells the
            // next call to run this pass with few workgroups, see qoimi_encode_batch)
            if (lane == 0 && atomicOr(&p.need_generic[img], 1u) == 0u) { if (atomicAdd(p.any_generic, 1u) == 0u) leave_hint(p); }
            return;
        }
    } else if (ENTRY == 3) {
        // ONE pass for every kind of content (experiment, QOIMI_ENC_UNI=1): a set whose look-back window determines its entry state takes it
        // from there as ever and leaves the table as it stands BEHIND it as INCLUSIVE granules; a set whose window does not (flat stretches)
        // walks its own pixels, publishes, looks back over the sets in front of it (g2_entry_state) - no image is flagged, no second launch.
        if (!warm_entry_state<CH, PROBE>(pix, lo, lane, L, C.tbase, in, last_edge)) {
            if (PROBE == 1) {
                if (!g2_entry_state<CH>(p, pix, n, lo, hi, p.g2_rec + (size_t)I.set_base * 65u, set, lane, L, C.tbase, last_edge, run_only)) return;
            }
            fell_back = true;
            if (lane == 0 && __hip_atomic_load(p.any_generic, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u && atomicOr(p.any_generic, 1u) == 0u) leave_hint(p);
        }
    } else if (ENTRY == 2) {
        if (PROBE == 1) {
            bool front = false;
            if (!g2_entry_state<CH>(p, pix, n, lo, hi, p.g2_rec + (size_t)I.set_base * 65u, set, lane, L, C.tbase, last_edge, run_only, &front)) return;
            // every image of the call by state look-back (no first pass: the context's previous batch held flagged images only): the count
            // of images with flat stretches is kept all the same - it decides how the NEXT batch runs
            if (p.all_g2 && front && lane == 0 && atomicOr(&p.need_generic[img], 1u) == 0u) atomicAdd(p.any_generic, 1u);
        }
        if (nint && !run_only) load_group<CH>(pix, lo, lane, ax, av);       // (from the L2 / Infinity Cache: the set's own walk has just read them)
    } else {
        const u64 lv = uniform64(in.tab_valid);
        L.table[lane] = ((lv >> lane) & 1ull) ? in.tab_loc : in.tab_far;
        last_edge = max(__builtin_amdgcn_readfirstlane(in.le_loc), __builtin_amdgcn_readfirstlane(in.le_far));   // max edge position < lo, or -1
    }
    if (PROBE == 0) L.mask[lane] = 0;
    // A set that begins BEFORE the image's first edge - the image opens with more pixels of the start value {0,0,0,255} (qoi.h:396-399)
    // than the sets in front of this one hold: a letterboxed frame's black rows - takes the general form too: its repeat pixels repeat
    // a value that no edge has written to the table, and the all-lanes probe of the plain form (probe_swap_all) would write it
    // there; the first later edge pixel of that value then found itself and became QOI_OP_INDEX 53 where qoi.h:430-436 finds the
    // zeroed slot and writes a literal chunk (round trip exact, bytes not the reference's; found by tests/fuzz_encode.py in round 4:
    // until then only the image's first set took the general form).  The group loaded ahead is dropped.
    if (last_edge < 0) nint = 0u;
    uint32_t ccp = (uint32_t)__builtin_amdgcn_readfirstlane((int)(63u + (uint32_t)((int)lo - last_edge)));
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_setprio(0);
    PHASE_MARK(0);

    const uint32_t sbase = lds_addr(L.stage);
    uint32_t vbase = sbase;                                    // LDS address of the next staged byte (same in every lane)
    uint32_t spilled = 0;                                      // bytes of the set already moved to its scratch slot
    uint32_t slot_id = 0xFFFFFFFFu;                            // pool mode: the slot the set holds (none yet)
    uint8_t* slot = p.pool ? nullptr : p.scratch + sg * p.set_stride;
    auto need_slot = [&]() {
        if (p.pool && slot_id == 0xFFFFFFFFu) {
            slot_id = pool_take(p, (uint32_t)sg * 2654435761u, lane);
            slot = p.scratch + (size_t)slot_id * p.set_stride;
        }
    };

    // Look-back, first poll: the records of the 64 sets before this one are asked for when the set's LAST group begins - the sets
    // before it started earlier and have mostly published by then - so that the answer travels while that group is encoded
    // (asked for after it, every set sat idle in its slot for the round trip: 10 % of the kernel).
    constexpr u64 kRecAgg = 1ull << 62, kRecIncl = 2ull << 62;
    const u64* const rec_base = p.status + (sg - set);        // look-back record of this image's set 0
    u64 early_rec = kRecIncl;
    bool early = false;
    auto ask_early = [&]() {
        if (p.lookback == 1 && set != 0u) { early_rec = lane < set ? granule_load(&rec_base[set - 1u - lane]) : kRecIncl; early = true; }
    };

    uint32_t g = 0;
    if ((ENTRY == 2 || ENTRY == 3) && run_only) {
        // Every pixel of the set repeats the pixel in front of it (constant frames, letterbox bars, blank pages): pixel i, d pixels behind
        // the last edge, carries 0xFD where d is a multiple of 62 (qoi.h:417-421) and nothing else - but the set's last pixel, which
        // closes its run if the image ends with it or the pixel behind it is an edge (qoi.h:425-428).  No second walk over the pixels.
        const uint32_t d0 = lo - (uint32_t)last_edge;                      // distance of the set's first pixel (last_edge = -1: the start value in front of pixel 0)
        const uint32_t d1 = d0 + (hi - lo) - 1u;                           // ... of its last one
        const uint32_t k = d1 / 62u - (d0 - 1u) / 62u;                     // multiples of 62 in [d0, d1]
        bool closes = last_set;
        if (!last_set) closes = load_px<CH>(pix, hi) != load_px<CH>(pix, hi - 1u);
        closes = closes && (d1 % 62u) != 0u;
        lds_u8* const st8 = (lds_u8*)(uintptr_t)sbase;
        for (uint32_t i = lane; i < k; i += 64u) st8[i] = (uint8_t)0xFDu;
        if (lane == 0 && closes) st8[k] = (uint8_t)(0xC0u | ((d1 - 1u) % 62u));
        vbase = sbase + k + (closes ? 1u : 0u);
        __builtin_amdgcn_wave_barrier();
        g = ngroups; nint = 0u;
        ask_early();
    }
    if (nint) {
        u64 E;
        { uint32_t p0, v0; unpack_pair<CH>(ax[0], av[0], p0, v0); E = __ballot(p0 != v0); }
        // ---- two groups per turn: while one is encoded the loads of the next are in flight ----------------------------
        // (the pair loaded when no group follows inside the loop: the two pixels around the end of the set, or around the
        // start of the image's last group - every lane reads the same two, only "is the next pixel an edge" is taken from them)
#pragma unroll 1
        for (;;) {
            {   // group g sits in a*; fetch g+1 into b*
                const uint32_t base = lo + g * kGroupPx;
                const uint32_t spos = (uint32_t)__builtin_amdgcn_readfirstlane((int)vbase) - sbase;
                if (spos > LDS::kSpill) { need_slot(); vbase = sbase + spill_stage<PROBE>(L, slot, spilled, spos, false, lane); }
                if (g + 1u < nint) load_group<CH>(pix, base + kGroupPx, lane, bx, bv);
                else { pack_pair<CH>(load_px<CH>(pix, base + kGroupPx), load_px<CH>(pix, base + kGroupPx - 1u), bx[0], bv[0]); if (!last_set) ask_early(); }
                process_group<PROBE, false, CH>(L, C, lane, ax, av, bx[0], bv[0], 0, E, ccp, vbase);
                if (++g >= nint) break;
            }
            {   // group g sits in b*; fetch g+1 into a*
                const uint32_t base = lo + g * kGroupPx;
                const uint32_t spos = (uint32_t)__builtin_amdgcn_readfirstlane((int)vbase) - sbase;
                if (spos > LDS::kSpill) { need_slot(); vbase = sbase + spill_stage<PROBE>(L, slot, spilled, spos, false, lane); }
                if (g + 1u < nint) load_group<CH>(pix, base + kGroupPx, lane, ax, av);
                else { pack_pair<CH>(load_px<CH>(pix, base + kGroupPx), load_px<CH>(pix, base + kGroupPx - 1u), ax[0], av[0]); if (!last_set) ask_early(); }
                process_group<PROBE, false, CH>(L, C, lane, bx, bv, ax[0], av[0], 0, E, ccp, vbase);
                if (++g >= nint) break;
            }
        }
    }
    PHASE_MARK(1);
    // ---- general form, group by group (no loads in flight across groups: one set per image, and one group more) -----------
#pragma unroll 1
    for (; g < ngroups; ++g) {
        const uint32_t base = lo + g * kGroupPx;
        const uint32_t spos = (uint32_t)__builtin_amdgcn_readfirstlane((int)vbase) - sbase;
        if (spos > LDS::kSpill) { need_slot(); vbase = sbase + spill_stage<PROBE>(L, slot, spilled, spos, false, lane); }
        uint32_t nxp, nxv;
#pragma unroll
        for (int t = 0; t < kGroupSteps; ++t) load_pair_guarded<CH>(pix, base + t * 64u + lane, n, ax[t], av[t]);
        load_pair_guarded<CH>(pix, base + kGroupPx + lane, n, nxp, nxv);
        if (g + 1u == ngroups) ask_early();
        u64 E = __ballot(ax[0] != av[0]);
        process_group<PROBE, true, 0>(L, C, lane, ax, av, nxp, nxv, (int)(n - base), E, ccp, vbase);
    }
    uint32_t spos = (uint32_t)__builtin_amdgcn_readfirstlane((int)vbase) - sbase;
    const uint32_t set_bytes = spilled + spos;
    PHASE_MARK(2);

    if (ENTRY == 3 && PROBE == 1 && !fell_back) {
        // the table as it stands behind this set, for a set further on whose look-back window will not do: every slot is known here (the
        // window filled all 64, or the image's start is in reach), the last edge follows from the distance counter
        __builtin_amdgcn_wave_barrier();
        const uint32_t tw = L.table[lane];
        const int le_end = (int)hi + 63 - (int)(uint32_t)__builtin_amdgcn_readfirstlane((int)ccp);
        u64* const mine = p.g2_rec + ((size_t)I.set_base + set) * 65u;
        const u64 tag = g2_tag(p);
        granule_store(&mine[lane], (u64)tw | (1ull << 32) | tag | kG2Incl);
        if (lane == 0) granule_store(&mine[64], (u64)(uint32_t)(le_end + 1) | tag | kG2Incl);
    }
    if (PIPE) {
        // the NEXT set of this wavefront: its ticket and its first loads go out here, in front of this set's placement
        if (has_next && __hip_atomic_load((gu32*)&p.need_generic[next_img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(&p.ticket[next_img], 1u);
            const uint32_t nset = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            const uint32_t nlo = nset * p.set_px;
            pre->img = next_img; pre->set = nset; pre->have_ticket = true;
            // (loads only for sets that begin inside the image with a whole first group in front of its end: the others - an image's
            // first and last set, tickets beyond its sets - do their own loads, or nothing, when their turn comes)
            if (nset < p.sets_per_image && nlo != 0u && nlo + kGroupPx < p.npx) {
                const uint8_t* __restrict__ npix = p.pixels + (size_t)next_img * p.pixel_stride;
                const uint8_t* __restrict__ q = npix + (size_t)(nlo + lane) * (size_t)CH;
#pragma unroll
                for (int k = 0; k < 8; ++k) load_pair_at<CH>(q, -64 * (k + 1), pre->warm[k], pre->warm_prev[k]);
                pre->valid = true;
            }
        }
    }
    if (!p.lookback) {
        // ---- order-free mode: park the set's bytes in its scratch slot, E4 (enc_offsets + enc_compact) places them ----
        (void)spill_stage<PROBE>(L, slot, spilled, spos, true, lane);
        if (lane == 0) p.set_size[sg] = set_bytes;
        return;
    }

    // ---- set byte count -> offset: decoupled look-back over the earlier sets of the image -----------------
    // A record is an 8-byte granule: bits 62..63 = 0 nothing yet, 1 the set's own byte count, 2 the byte count of the image up
    // to and including the set; the count sits in the low dword (a stream is shorter than 2^31 bytes, qoi.h:328-332).
    uint32_t excl = 0;
    if (p.lookback == 2) {
        // ---- tree placement (calls of a few large images): three levels of byte counts, 64 to a window -------------------------
        // With ONE image every wavefront in flight belongs to it - thousands of sets that finish at about the same time - and the
        // inclusive prefixes of the look-back below travel 64 sets per round trip through them.  Here nothing travels: a set
        // publishes its own count; the LAST set of a group of 64 adds the counts of its group and publishes the group's total, the
        // last set of 64 groups the total of those; and every set adds, at once, the counts before it in its group, the group totals
        // before its group in its block, and the block totals before its block - three loads per poll, depth three whatever the
        // image's size (64^3 sets: 800 Mpx at one slab per set).  A wait is for lower-numbered sets only; units go to the workgroups in
        // START order (enc_sets: one ticket per workgroup), so those are resident or done (the spin bound still guards it).
        const uint32_t nsets = I.sets;
        const uint32_t n1 = (nsets + 63u) >> 6, n2 = (n1 + 63u) >> 6;
        u64* const t1 = p.tree1 + (size_t)img * n1;
        u64* const t2 = p.tree2 + (size_t)img * n2;
        const uint32_t g1 = set >> 6, j0 = set & 63u, g2 = g1 >> 6, j1 = g1 & 63u;
        if (lane == 0) granule_store(&p.status[sg], kRecAgg | set_bytes);
        const bool close1 = j0 == 63u || set + 1u == nsets;                // this set publishes its group's total ...
        const bool close2 = close1 && (j1 == 63u || g1 + 1u == n1);        // ... and its block's
        bool d0 = j0 == 0u, d1 = j1 == 0u, d2 = g2 == 0u, pub1 = !close1, pub2 = !close2;
        uint32_t a0 = 0, a1 = 0, a2 = 0, spins = 0;
        while (!(d0 && d1 && d2)) {
            u64 v0 = kRecAgg, v1 = kRecAgg, v2 = kRecAgg;
            if (!d0 && lane < j0) v0 = granule_load(&rec_base[(g1 << 6) + lane]);
            if (!d1 && lane < j1) v1 = granule_load(&t1[(g2 << 6) + lane]);
            if (!d2 && lane < g2) v2 = granule_load(&t2[lane]);
            if (!d0 && lanes_where((uint32_t)(v0 >> 62) == 0u) == 0ull) { a0 = wave_sum32_upto((uint32_t)v0, lane, 63); d0 = true; }
            if (!d1 && lanes_where((uint32_t)(v1 >> 62) == 0u) == 0ull) { a1 = wave_sum32_upto((uint32_t)v1, lane, 63); d1 = true; }
            if (!d2 && lanes_where((uint32_t)(v2 >> 62) == 0u) == 0ull) { a2 = wave_sum32_upto((uint32_t)v2, lane, 63); d2 = true; }
            if (d0 && !pub1) { if (lane == 0) granule_store(&t1[g1], kRecAgg | (u64)(a0 + set_bytes)); pub1 = true; }
            if (d0 && d1 && !pub2) { if (lane == 0) granule_store(&t2[g2], kRecAgg | (u64)(a1 + a0 + set_bytes)); pub2 = true; }
            if (d0 && d1 && d2) break;
            if (ENTRY == 1 && (spins & 7u) == 7u && __hip_atomic_load((gu32*)&p.need_generic[img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                if (slot_id != 0xFFFFFFFFu) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pool_give(p, slot_id, lane); }
                return;
            }
            // A tripped bound (never observed) is published at once and ends every wait of the launch: the waiters look at the flag every
            // 64th poll and leave WITHOUT copying out (their offsets would be wrong); the host re-encodes such a call order-free.
            if (++spins > p.spin_bound || ((spins & 63u) == 63u && __hip_atomic_load((gu32*)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                if (lane == 0) atomicOr(p.err, 1u);
                if (slot_id != 0xFFFFFFFFu) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pool_give(p, slot_id, lane); }
                return;
            }
            __builtin_amdgcn_s_sleep(2);
        }
        excl = a0 + a1 + a2;
    } else {
        u64* st = p.status;
        if (set == 0) {
            if (lane == 0) granule_store(&st[sg], kRecIncl | set_bytes);
        } else {
            if (lane == 0) granule_store(&st[sg], kRecAgg | set_bytes);
            uint32_t look = set - 1u;                          // newest set of the current window (index within the image)
            uint32_t spins = 0;
            for (;;) {
                const bool inwin = lane <= look;
                u64 v;
                if (early) { v = early_rec; early = false; }               // the answer to ask_early(): the same window
                else v = inwin ? granule_load(&rec_base[look - lane]) : kRecIncl;   // before set 0: inclusive prefix 0
                const uint32_t flag = (uint32_t)(v >> 62);
                const u64 notready = __ballot(flag == 0u);
                const u64 incl = __ballot(flag == 2u);
                const int stop = incl ? __builtin_ctzll(incl) : 64;       // nearest inclusive record
                const u64 need = stop >= 64 ? ~0ull : ((1ull << stop) - 1ull);
                if (notready & need) {                                    // a record we must add is not published yet
                    // (first pass: a predecessor that gave the image up never publishes - the image is encoded again anyway.  Looked at
                    // every 8th poll only: the flag is one more round trip through the fabric in front of every re-poll)
                    if (ENTRY == 1 && (spins & 7u) == 7u && __hip_atomic_load((gu32*)&p.need_generic[img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
                        if (slot_id != 0xFFFFFFFFu) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pool_give(p, slot_id, lane); }
                        return;
                    }
                    if (++spins > p.spin_bound || ((spins & 63u) == 63u && __hip_atomic_load((gu32*)p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u)) {
                        if (lane == 0) atomicOr(p.err, 1u);           // (see the tree's wait above)
                        if (slot_id != 0xFFFFFFFFu) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); pool_give(p, slot_id, lane); }
                        return;
                    }
                    __builtin_amdgcn_s_sleep(2);
                    continue;
                }
                excl += wave_sum32_upto((uint32_t)v, lane, stop);
                if (incl) break;
                look -= 64u;                                              // (no inclusive record among 64: look >= 64 here)
            }
            if (lane == 0) granule_store(&st[sg], kRecIncl | (u64)(excl + set_bytes));
#ifdef QOIMI_ENC_PHASES
            if (lane == 0) { atomicAdd(&g_enc_phase[7], (unsigned long long)spins); if (spins) atomicAdd(&g_enc_phase[6], 1ull); }
#endif
        }
    }

    PHASE_MARK(3);
    // ---- copy the set's bytes out ----------------------------------------------------------
    uint8_t* __restrict__ out = p.out + I.out_off;
    if (set == 0 && lane < (uint32_t)kHeaderBytes) {        // 14-byte header (qoi.h:384-388)
        const uint32_t w = I.width, h = I.height;
        const u64 hdr_lo = 0x66696F71ull | ((u64)__builtin_bswap32(w) << 32);             // "qoif", width BE
        const u64 hdr_hi = (u64)__builtin_bswap32(h) | ((u64)p.channels << 32) | ((u64)I.colorspace << 40);
        out[lane] = (uint8_t)((lane < 8u ? hdr_lo : hdr_hi) >> (8u * (lane & 7u)));
    }
    const u64 pos = (u64)kHeaderBytes + (u64)excl;
    if (spilled) {                                          // the part that went through the scratch slot: by this wavefront, from this CU
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        copy_global_out(slot, out + pos, spilled, lane);
        if (slot_id != 0xFFFFFFFFu) pool_give(p, slot_id, lane);
    }
    if (spos) {
        __builtin_amdgcn_wave_barrier();
        copy_stage_out(L.stage, out + pos + spilled, spos, lane);
    }
    if (last_set) {                                         // trailer (qoi.h:339,480-482) + *out_len
        const u64 end = pos + set_bytes;
        if (lane < (uint32_t)kTrailerBytes) out[end + lane] = (lane == 7u) ? 1 : 0;
        if (lane == 0) p.out_len[I.len_index] = (int)(end + kTrailerBytes);
    }
#ifdef QOIMI_ENC_PHASES
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    PHASE_MARK(4);
}

// ENTRY 0: entry state from the per-slab summaries + scans (E1/E2).  ENTRY 1: each set finds it itself
// (warm_entry_state); sets whose look-back window does not suffice flag their image, and the launcher runs
// the ENTRY 0 passes with only_flagged set: small grid-stride grids that return at once when nothing was
// flagged.  A workgroup serves unit u = (image u % n_images, four consecutive sets u / n_images) so that the
// sets in flight spread over all images.
template <int CH, int PROBE, int ENTRY, bool MIXED, bool PIPE = false>
// (the generic 3-channel form - flat 3-channel images only - takes a register more than six wavefronts per SIMD leave it: five)
// (and so do the forms for differently shaped images, whose geometry comes from a table)
__global__ __launch_bounds__(256, PROBE == 1 ? ((CH == 3 && ENTRY == 0) || MIXED ? QOIMI_ENC_WAVES_PER_SIMD - 1 : QOIMI_ENC_WAVES_PER_SIMD) : 4) void enc_sets(EncParams p) {
    __shared__ EncLdsFor<PROBE> s_lds[4];
    __shared__ uint32_t s_unit;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    if (p.only_flagged && *p.any_generic == 0u) return;
    // One frame: the hipMemsetAsync over the call's records, tickets, flags and pool map was a tenth of the call (2.9 us + a launch gap of
    // 35).  A context keeps TWO such regions for calls of a few images and uses them in turn; the first launch of a call zeroes the other
    // one in passing - a store per thread of its first few workgroups - and the next call finds it zeroed (qoimi_encode_batch checks that
    // it is the very next call and lays its region out the same way; anything else pays the memset as before).
    if ((ENTRY == 1 || ENTRY == 3) && p.zero_next != nullptr)
        for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < p.zero_next_dwords; i += gridDim.x * 256u) p.zero_next[i] = 0u;
    SetPre pre;
    pre.have_ticket = false; pre.valid = false;
#pragma unroll 1
    for (uint32_t unit0 = blockIdx.x; unit0 < p.n_units; unit0 += gridDim.x) {
        uint32_t unit = unit0;
        if (PIPE) {
            // (look-back placement with tickets and SPREAD, one shape: what the launcher selects this form for)
            const uint32_t img = (unit * 4u + wave) % p.n_images;
            const bool mine = pre.have_ticket, loaded = pre.valid;         // taken / asked for by this wavefront's previous set
            pre.have_ticket = false; pre.valid = false;
            uint32_t set;
            if (mine) set = pre.set;
            else {
                if (__hip_atomic_load((gu32*)&p.need_generic[img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) continue;
                uint32_t t = 0;
                if (lane == 0) t = atomicAdd(&p.ticket[img], 1u);
                set = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
            }
            const uint32_t unit2 = unit0 + gridDim.x;
            if (set < p.sets_per_image && __hip_atomic_load((gu32*)&p.need_generic[img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u)
                encode_set<CH, PROBE, ENTRY, MIXED, true>(p, img, set, lane, s_lds[wave], &pre, loaded, unit2 < p.n_units, (unit2 * 4u + wave) % p.n_images);
            __builtin_amdgcn_wave_barrier();
            continue;
        }
        if (p.lookback == 2 && p.use_ticket) {
            // Tree placement waits for LOWER-numbered sets of the same launch.  Units handed out by workgroup index would make that a bet
            // on in-order dispatch (every XCD dispatches its share on its own: two launches from different streams could wait on each
            // other in a circle).  One ticket per WORKGROUP (a quarter of the per-wavefront tickets that cost a lone image 60 us,
            // EXPERIMENTS.md): units in START order - whatever a set waits for is resident or done.  (One pass of this loop: the
            // launcher gives tree calls one workgroup per unit.)
            if (threadIdx.x == 0) s_unit = atomicAdd(&p.ticket[0], 1u);
            __syncthreads();
            unit = s_unit;
        }
        // p.spread (look-back + ticket mode; the default): the four wavefronts of a workgroup serve four consecutive IMAGES instead of
        // taking four consecutive tickets of one image at the same instant.  An image's consecutive tickets then go to wavefronts that
        // started at different times (1024 x 4K photographs: 12.45 -> 12.23 ms, profiles/r04_s1_enc_knobs.txt).
        // Every image still receives sets_per_image tickets' worth of wavefronts (4 n_units / n_images of them).
        uint32_t img, set, sets_of_img = p.sets_per_image;
        if (MIXED) {
            // differently shaped images (order-free placement only): the units of an image follow each other
            img = find_by_unit_base(p, unit);
            set = (unit - p.img_tab[img].unit_base) * 4u + wave;
            sets_of_img = p.img_tab[img].sets;
        } else {
            img = (p.spread && p.use_ticket && p.lookback == 1) ? (unit * 4u + wave) % p.n_images : unit % p.n_images;
            set = (unit / p.n_images) * 4u + wave;         // order-free mode: any order will do
        }
        if (p.only_flagged && p.need_generic[img] == 0u) continue;
        if (ENTRY == 1 && __hip_atomic_load((gu32*)&p.need_generic[img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) continue;   // image already sent to the generic path
        if (p.use_ticket && p.lookback == 1) {
            // look-back mode: the sets of an image are handed out by the image's ticket counter, one ticket per WAVEFRONT, i.e.
            // in START order: every predecessor a look-back can wait on is already running or finished (no reliance on
            // dispatch order; guide G16).  One counter per image keeps the atomics off a single hot word; no workgroup barrier,
            // so a wavefront that waits in its look-back does not hold up the other three of its workgroup.
            uint32_t t = 0;
            if (lane == 0) t = atomicAdd(&p.ticket[img], 1u);
            set = (uint32_t)__builtin_amdgcn_readfirstlane((int)t);
        }
        if (set < sets_of_img) encode_set<CH, PROBE, ENTRY, MIXED>(p, img, set, lane, s_lds[wave]);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------
// E4a (order-free mode): exclusive scan of the set byte counts of one image (one workgroup per image);
// also writes the 14-byte header (qoi.h:384-388), the 8-byte end marker (qoi.h:339,480-482)
// and *out_len (qoi.h:484).
// ---------------------------------------------------------------------------------
// The scan walks the image in tiles of 16384 sets, 1024 per wavefront: counts are loaded coalesced (next tile's
// while this one is scanned), turned through a wavefront-private LDS stripe so that a lane holds 16 consecutive
// counts, scanned (lane-serial, then six rounds over the wavefront, then over the 16 wavefronts) and written back
// the same way.
template <bool MIXED>
__global__ __launch_bounds__(1024) void enc_offsets(EncParams p) {
    constexpr uint32_t kPer = 16, kStripe = 64u * kPer, kTile = 16u * kStripe;
    __shared__ uint32_t s_turn[16][kStripe + 64u];             // element e of a stripe at e + e/16 (bank spread)
    __shared__ uint32_t s_wave[16];
    const uint32_t img = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (p.only_flagged && (*p.any_generic == 0u || p.need_generic[img] == 0u)) return;     // placement of the generic pass's images only
    const EncImage I = enc_image<MIXED>(p, img);
    const uint32_t* __restrict__ sz = p.set_size + (size_t)I.set_base;
    uint32_t* __restrict__ off = p.set_off + (size_t)I.set_base;
    const uint32_t n = I.sets;
    uint32_t* turn = s_turn[wave];
    uint32_t carry = 0;
    uint32_t nv[kPer];
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) { const uint32_t e = wave * kStripe + j * 64u + lane; nv[j] = e < n ? sz[e] : 0u; }
    for (uint32_t base = 0; base < n; base += kTile) {
        const uint32_t sbase = base + wave * kStripe;          // first set of this wavefront's stripe
#pragma unroll
        for (uint32_t j = 0; j < kPer; ++j) { const uint32_t e = j * 64u + lane; turn[e + (e >> 4)] = nv[j]; }
#pragma unroll
        for (uint32_t j = 0; j < kPer; ++j) { const uint32_t e = sbase + kTile + j * 64u + lane; nv[j] = e < n ? sz[e] : 0u; }   // next tile
        __builtin_amdgcn_wave_barrier();
        uint32_t v[kPer];
        uint32_t mine = 0;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) { v[k] = turn[lane * 17u + k]; mine += v[k]; }
        uint32_t incl = mine;                                  // inclusive scan over the wavefront
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint32_t up = gather_lane(incl, lane - d);
            if (lane >= d) incl += up;
        }
        if (lane == 63u) s_wave[wave] = incl;
        __syncthreads();
        uint32_t before = 0, total = 0;
#pragma unroll
        for (uint32_t k = 0; k < 16u; ++k) { const uint32_t s = s_wave[k]; before += k < wave ? s : 0u; total += s; }
        __syncthreads();                                       // s_wave is rewritten by the next tile
        uint32_t run = carry + before + incl - mine;
#pragma unroll
        for (uint32_t k = 0; k < kPer; ++k) { turn[lane * 17u + k] = run; run += v[k]; }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (uint32_t j = 0; j < kPer; ++j) { const uint32_t e = j * 64u + lane; if (sbase + e < n) off[sbase + e] = turn[e + (e >> 4)]; }
        __builtin_amdgcn_wave_barrier();
        carry += total;
    }
    uint8_t* out = p.out + I.out_off;
    const uint32_t total = carry;
    if (tid < (uint32_t)kHeaderBytes) {
        const uint32_t w = I.width, h = I.height;
        const u64 hdr_lo = 0x66696F71ull | ((u64)__builtin_bswap32(w) << 32);             // "qoif", width BE
        const u64 hdr_hi = (u64)__builtin_bswap32(h) | ((u64)p.channels << 32) | ((u64)I.colorspace << 40);
        out[tid] = (uint8_t)((tid < 8u ? hdr_lo : hdr_hi) >> (8u * (tid & 7u)));
    }
    if (tid < (uint32_t)kTrailerBytes) out[(size_t)kHeaderBytes + total + tid] = (tid == 7u) ? 1 : 0;
    if (tid == 0) p.out_len[I.len_index] = (int)(kHeaderBytes + total + kTrailerBytes);
}

// E4b (order-free mode): move every set's bytes from its scratch slot to its place in the stream
// (one wavefront per set; aligned 16-byte stores, source re-aligned with v_alignbyte).
template <bool MIXED>
__global__ __launch_bounds__(256) void enc_compact(EncParams p) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    if (p.only_flagged && *p.any_generic == 0u) return;
    const size_t total = MIXED ? (size_t)p.img_tab[p.n_images].set_base : (size_t)p.n_images * p.sets_per_image;
#pragma unroll 1
    for (size_t sg = (size_t)blockIdx.x * 4u + wave; sg < total; sg += (size_t)gridDim.x * 4u) {
        const uint32_t img = MIXED ? find_by_set_base(p, (uint32_t)sg) : (uint32_t)(sg / p.sets_per_image);
        if (p.only_flagged && p.need_generic[img] == 0u) continue;
        const uint32_t n = p.set_size[sg];
        if (n == 0) continue;
        const size_t out_off = MIXED ? p.img_tab[img].out_off : (size_t)img * p.out_stride;
        copy_global_out(p.scratch + sg * p.set_stride, p.out + out_off + kHeaderBytes + p.set_off[sg], n, lane);
    }
}

// Measures whether one ds_wrxchg_rtn_b32 serves same-address lanes in ascending lane order
// (see PROBE above).  out[0] = number of mismatching patterns (0: PROBE 1 is usable).
// What PROBE 1 rests on is a measured property, not a documented one, so it is measured where it matters: alone and under
// contention at context creation, and again every 256 encode calls of a context while it runs (qoi_host.hip; the result is
// read at the following call: a failure switches the context to the order-free probe and reports the call as failed).
// (An in-kernel cross-check of one slab in 64 against the order-free rule was tried in round 2: its extra instantiation
// cost the whole kernel 28 bytes of scratch per lane and 5 % of its time.)
// WAVES wavefronts per workgroup run the test at the same time, each on its own 64 words: with WAVES = 4 and a grid that
// fills every CU sixteen of them hammer one LDS at once (the stress variant the round-1 review asked for).
template <int WAVES>
__global__ __launch_bounds__(64 * WAVES) void lds_order_selftest(uint32_t* out) {
    __shared__ uint32_t tabs[WAVES][64];
    uint32_t* tab = tabs[threadIdx.x >> 6];
    const uint32_t lane = lane_id();
    uint32_t bad = 0;
    uint32_t rng = 0x9E3779B9u * (blockIdx.x * WAVES + (threadIdx.x >> 6) + 1u) + lane * 0x85EBCA6Bu;
    for (int it = 0; it < (WAVES == 1 ? 256 : 48); ++it) {
        rng = rng * 1664525u + 1013904223u;
        const uint32_t nb = 1u << ((it % 7));                          // 1..64 distinct slots
        const uint32_t slot = ((rng >> 16) % nb) * (64u / nb);
        const bool on = ((rng >> 8) & 7u) != 0u || nb == 1u;
        tab[lane] = 0xFFFF0000u | lane;
        __builtin_amdgcn_wave_barrier();
        const uint32_t val = (uint32_t)it * 64u + lane;
        uint32_t old = 0;
        if (on) old = __hip_atomic_exchange(&tab[slot], val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
        __builtin_amdgcn_wave_barrier();
        // model: nearest lower active lane with the same slot, else the initial content
        const u64 act = __ballot(on);
        u64 same = 0;
        for (uint32_t l = 0; l < 64u; ++l) {
            const uint32_t sl = read_lane_dyn(slot, l);
            if (sl == slot && ((act >> l) & 1ull)) same |= 1ull << l;
        }
        const u64 pred = same & ((1ull << lane) - 1ull);
        const uint32_t want = pred ? (uint32_t)it * 64u + (uint32_t)msb64(pred) : (0xFFFF0000u | slot);
        if (on && old != want) ++bad;
        // final content: highest active lane per slot
        const uint32_t fin = tab[lane];
        u64 mine = 0;
        for (uint32_t l = 0; l < 64u; ++l) {
            const uint32_t sl = read_lane_dyn(slot, l);
            if (sl == lane && ((act >> l) & 1ull)) mine |= 1ull << l;
        }
        const uint32_t wantf = mine ? (uint32_t)it * 64u + (uint32_t)msb64(mine) : (0xFFFF0000u | lane);
        if (fin != wantf) ++bad;
        __builtin_amdgcn_wave_barrier();
    }
    bad = wave_sum(bad);
    if (lane == 0 && bad) atomicAdd(out, bad);
}

// ---------------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------------
template <int CH, int PROBE, bool MIXED>
static void launch_encode_t(EncParams p, hipStream_t st, KernelTimer* tm, int phases, uint32_t mixed_units, uint32_t mixed_slabs, uint32_t mixed_groups, uint32_t mixed_sets) {
    const uint32_t total_slabs = MIXED ? mixed_slabs : p.n_images * p.spi;
    const uint32_t total_groups = MIXED ? mixed_groups : p.n_images * p.gpi;
    const uint32_t total_sets = MIXED ? mixed_sets : p.n_images * p.sets_per_image;
    p.total_slabs = total_slabs;
    const uint32_t slab_blocks = (total_slabs + 3u) / 4u;
    const uint32_t quads_per_image = (p.sets_per_image + 3u) / 4u;
    p.n_units = MIXED ? mixed_units : quads_per_image * p.n_images;
    const bool warm = p.warm && PROBE == 1;
    // grid of the passes that usually have nothing to do: they return at once then (a few microseconds for 2048 workgroups).  When
    // there IS work - flat content - a grid-stride loop over few long-lived workgroups is the slow way to run enc_sets (a persistent
    // grid costs it 20 %, profiles/r04_s1_enc_knobs.txt): large calls get 1/32 of the full grid (256 flat 4K frames: 8.5 -> 7.3 ms;
    // 1/8: 7.0 ms, but its empty launches cost the photographs 0.7 %; 1/32: 0.2 %).
    // (round 6: 1/8 for calls of up to 128 K units, whose empty launch of up to 16 K workgroups is ~10 us: the mixed directory's qoimi_encode_images - a third of
    // its images flat - 1.39 -> 1.26 ms; 128 UI frames behind a batch of photographs stay at 3.1 ms against 2.8 primed: that is the first pass a primed call skips)
    const uint32_t most = p.n_units > slab_blocks ? p.n_units : slab_blocks;
    const uint32_t gen_div = p.gen_small_div ? p.gen_small_div : (most <= 131072u ? 8u : 32u);            // (QOIMI_ENC_GEN_GRID_DIV, under QOIMI_TUNING: qoimi_ctx_create)
    uint32_t small = 2048u;
    { const uint32_t big = most / gen_div; if (big > small) small = big; }
    tm->mark(kT_begin, st);
    if (phases & kEncSlabs) {
    if (warm) {
        p.only_flagged = 0;
        // (p.persist: test knob - at most that many workgroups, each taking unit after unit in the kernel's grid-stride loop)
        // (tree placement takes its sets by workgroup index and waits for lower-numbered ones: one workgroup per unit, no grid-stride loop)
        if (p.persist == 0xFFFFFFFFu) p.persist = p.n_units / 16u > 2048u ? p.n_units / 16u : 2048u;   // (the previous batch held flagged images only: qoimi_encode_batch)
        bool piped = false;
        if constexpr (!MIXED && PROBE == 1) {
            if (p.uni && p.lookback != 0 && p.g2_rec != nullptr) {          // one pass, a set falls back to the state look-back by itself
                hipLaunchKernelGGL((enc_sets<CH, PROBE, 3, MIXED>), dim3(p.n_units), dim3(256), 0, st, p);
                piped = true;
            }
        }
        if constexpr (!MIXED && PROBE == 1) {
            // experiment (QOIMI_ENC_PIPE=1 with QOIMI_ENC_PERSIST=N): each wavefront's next set asked for in front of its current set's placement
            if (!piped && p.pipe && p.persist && p.n_units > p.persist && p.lookback == 1 && p.use_ticket && p.spread) {
                hipLaunchKernelGGL((enc_sets<CH, PROBE, 1, MIXED, true>), dim3(p.persist), dim3(256), 0, st, p);
                piped = true;
            }
        }
        // (p.all_g2: no first pass at all - the pass over flagged images takes every image; qoimi_encode_batch)
        const bool skip_first = !MIXED && PROBE == 1 && p.all_g2 && p.lookback == 1 && p.g2_rec != nullptr && !p.uni;
        if (!piped && !skip_first)
        hipLaunchKernelGGL((enc_sets<CH, PROBE, 1, MIXED>), dim3(p.persist && p.n_units > p.persist && p.lookback != 2 ? p.persist : p.n_units), dim3(256), 0, st, p);
        tm->mark(kT_enc_slabs, st);
        p.only_flagged = skip_first ? 0 : 1;
    } else {
        p.only_flagged = 0;
        small = 0xFFFFFFFFu;
    }
    const bool first_lookback = p.lookback != 0;
    if constexpr (!MIXED && PROBE == 1) if (warm && first_lookback && p.g2_rec != nullptr && p.uni) return;       // ENTRY 3 did it all
    if constexpr (!MIXED && PROBE == 1) if (warm && first_lookback && p.g2_rec != nullptr) {
        // The images the first pass gave up on (flat content), by state look-back over their sets (ENTRY 2, g2_entry_state): one launch
        // that returns at once when nothing was flagged - no summary pass, no scans.
        EncParams g = p;
        g.status = p.status_gen; g.ticket = p.ticket_gen; g.tree1 = p.tree1_gen; g.tree2 = p.tree2_gen;
        g.set_slabs = p.gen_slabs; g.set_px = p.gen_slabs * kEncSlabPx;
        g.sets_per_image = (p.spi + p.gen_slabs - 1u) / p.gen_slabs;
        g.n_units = ((g.sets_per_image + 3u) / 4u) * p.n_images;
        // (grid: 1/32 of the units when the call is not expected to hold flagged images - the launch then finds nothing and costs what its
        // grid is - and 1/p.gen_grid_div of them when the context's previous batch did: a loop over few long-lived workgroups is the slow
        // way to run this kernel, EXPERIMENTS.md)
        uint32_t grid2 = small;
        if (p.gen_grid_div) { const uint32_t want = g.n_units / p.gen_grid_div; if (want > grid2) grid2 = want; }
        if (!p.only_flagged) grid2 = g.n_units;
        hipLaunchKernelGGL((enc_sets<CH, PROBE, 2, MIXED>), dim3(g.n_units < grid2 || g.lookback == 2 ? g.n_units : grid2), dim3(256), 0, st, g);
        tm->mark(kT_enc_slabs_generic, st);
        return;
    }
    hipLaunchKernelGGL((enc_slab_summary<CH, kEncSteps, MIXED>), dim3(slab_blocks < small ? slab_blocks : small), dim3(256), 0, st, p);
    tm->mark(kT_enc_summary, st);
    hipLaunchKernelGGL(enc_scan_groups<MIXED>, dim3(total_groups), dim3(64), 0, st, p);
    tm->mark(kT_enc_scan_groups, st);
    hipLaunchKernelGGL(enc_scan_images<MIXED>, dim3(p.n_images), dim3(64), 0, st, p);
    tm->mark(kT_enc_scan_images, st);
    // The images the first pass gave up on (flat content) are encoded again from their first set ...
    EncParams g = p;
    if (warm && first_lookback) {
        // ... by look-back as well, with records / tickets of their own (the first pass left some behind for these images) and
        // kEncGenSetSlabs slabs per set: flat content is a few bytes per slab, eight slabs per look-back instead of R.  (Round 3 parked
        // these sets order-free in per-set scratch slots - the worst-case slots of EVERY set of the batch, 42.5 GB for 1024 4K frames.)
        g.status = p.status_gen; g.ticket = p.ticket_gen; g.tree1 = p.tree1_gen; g.tree2 = p.tree2_gen;
        g.set_slabs = p.gen_slabs; g.set_px = p.gen_slabs * kEncSlabPx;
        g.sets_per_image = (p.spi + p.gen_slabs - 1u) / p.gen_slabs;
        g.n_units = ((g.sets_per_image + 3u) / 4u) * p.n_images;
    } else if (warm) {
        g.lookback = 0; p.lookback = 0;                      // an order-free call: the flagged images are parked and placed with the others
    }
    hipLaunchKernelGGL((enc_sets<CH, PROBE, 0, MIXED>), dim3(g.n_units < small || g.lookback == 2 ? g.n_units : small), dim3(256), 0, st, g);
    tm->mark(warm ? kT_enc_slabs_generic : kT_enc_slabs, st);
    p.only_flagged = 0;
    }
    if (!p.lookback && (phases & kEncPlace)) {
        const uint32_t set_blocks = (total_sets + 3u) / 4u;
        hipLaunchKernelGGL(enc_offsets<MIXED>, dim3(p.n_images), dim3(1024), 0, st, p);
        tm->mark(kT_enc_offsets, st);
        hipLaunchKernelGGL(enc_compact<MIXED>, dim3(p.only_flagged && set_blocks > small ? small : set_blocks), dim3(256), 0, st, p);
        tm->mark(kT_enc_compact, st);
    }
}

void launch_encode(const EncParams& p, hipStream_t st, KernelTimer* tm, int phases) {
    if (p.channels == 3) {
        if (!p.probe_xchg) launch_encode_t<3, 0, false>(p, st, tm, phases, 0, 0, 0, 0);
        else launch_encode_t<3, 1, false>(p, st, tm, phases, 0, 0, 0, 0);
        return;
    }
    if (!p.probe_xchg) launch_encode_t<4, 0, false>(p, st, tm, phases, 0, 0, 0, 0);
    else launch_encode_t<4, 1, false>(p, st, tm, phases, 0, 0, 0, 0);
}
// differently shaped images (EncParams::img_tab; order-free placement): totals of the image table
void launch_encode_mixed(const EncParams& p, uint32_t units, uint32_t slabs, uint32_t groups, uint32_t sets, hipStream_t st, KernelTimer* tm) {
    if (p.channels == 3) {
        if (!p.probe_xchg) launch_encode_t<3, 0, true>(p, st, tm, kEncAll, units, slabs, groups, sets);
        else launch_encode_t<3, 1, true>(p, st, tm, kEncAll, units, slabs, groups, sets);
        return;
    }
    if (!p.probe_xchg) launch_encode_t<4, 0, true>(p, st, tm, kEncAll, units, slabs, groups, sets);
    else launch_encode_t<4, 1, true>(p, st, tm, kEncAll, units, slabs, groups, sets);
}

// returns the number of mismatching patterns of the LDS exchange-order self-test (0 = ordered)
// asynchronous form: zeroes *d_out and launches both variants on st; *d_out != 0 afterwards = the order does not hold
void launch_lds_order_selftest(uint32_t* d_out, hipStream_t st) {
    (void)hipMemsetAsync(d_out, 0, sizeof(uint32_t), st);
    hipLaunchKernelGGL(lds_order_selftest<1>, dim3(128), dim3(64), 0, st, d_out);
    hipLaunchKernelGGL(lds_order_selftest<4>, dim3(1024), dim3(256), 0, st, d_out);     // 48 patterns per wavefront: a few hundred microseconds
}

int run_lds_order_selftest(hipStream_t st) {
    uint32_t* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(uint32_t)) != hipSuccess) return -1;
    (void)hipMemsetAsync(d, 0, sizeof(uint32_t), st);
    hipLaunchKernelGGL(lds_order_selftest<1>, dim3(512), dim3(64), 0, st, d);          // one wavefront per workgroup
    hipLaunchKernelGGL(lds_order_selftest<4>, dim3(2048), dim3(256), 0, st, d);        // sixteen per CU at once (48 patterns each)
    uint32_t h = 1;
    if (hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) h = 0xFFFFFFFFu;
    (void)hipFree(d);
    return (int)h;
}

}  // namespace qoimi

#ifdef QOIMI_ENC_PHASES
extern "C" int qoimi_debug_enc_phases(unsigned long long* out, int reset) {
    if (hipDeviceSynchronize() != hipSuccess) return -1;
    if (out && hipMemcpyFromSymbol(out, HIP_SYMBOL(qoimi::g_enc_phase), sizeof(unsigned long long) * 8) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(qoimi::g_enc_phase), z, sizeof z) != hipSuccess) return -1; }
    return 0;
}
#endif
