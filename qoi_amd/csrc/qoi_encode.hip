// qoi_encode.hip — MI355X-native QOI encoder kernels (gfx950, wave64).
//
// Replaces the sequential loop of the reference encoder (qoi.h:356-486) with a
// slab-parallel evaluation of the SAME function of the input pixels; the emitted
// stream is byte-identical to the reference's (tests/test_gpu_encode.py).
//
// Every output byte is a pure function of the input:
//   * prev pixel           = the neighbouring input pixel (qoi.h:477)
//   * "edge" pixel         = px[i] != px[i-1] (qoi.h:415)
//   * run bytes            = function of the distance to the last edge (qoi.h:416-428)
//   * colour-table content = last edge pixel per hash slot (qoi.h:430-436)
// so an image is cut into slabs of 64*K pixels (one wavefront each) and only three
// small quantities are carried between slabs:
//   (1) the 64-entry colour table       -> passes E1 (slab summary) + E2 (scan)
//   (2) the position of the last edge   -> same passes
//   (3) the output byte offset          -> decoupled look-back inside pass E3
//
// Pass E3 is the hot kernel: one coalesced dword load per pixel, an LDS-resident
// colour table per wavefront, 64-bit ballots / mbcnt for run lengths and byte
// offsets.  Byte/integer work only - no MFMA.
#include "qoi_dev.h"
#include "qoi_kernels.h"

namespace qoimi {

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

__device__ __forceinline__ int msb64(u64 m) { return 63 - __builtin_clzll(m); }

// sum of v over lanes 0..stop (stop >= 63: all lanes)
__device__ __forceinline__ u64 wave_sum64_upto(u64 v, uint32_t lane, int stop) {
    u64 x = ((int)lane <= stop) ? v : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// ---------------------------------------------------------------------------------
// E1: per-slab summary.  For each hash slot the LAST edge pixel of the slab (+ valid
// bit) and the position of the slab's last edge (-1: none).   [qoi.h:415,430,436]
// One wavefront per slab; LDS ds_max_u64 on (position,value) keys keeps the latest.
// ---------------------------------------------------------------------------------
template <int CH, int K>
__global__ __launch_bounds__(256) void enc_slab_summary(EncParams p) {
    __shared__ u64 s_key[4][64];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    if (p.only_flagged && *p.any_generic == 0u) return;
    const uint32_t total = p.n_images * p.spi;
#pragma unroll 1
    for (uint32_t blk = blockIdx.x; blk * 4u < total; blk += gridDim.x) {
    const uint32_t g = blk * 4u + wave;
    if (g >= total) continue;
    const uint32_t img = g / p.spi, s = g - img * p.spi;
    if (p.only_flagged && p.need_generic[img] == 0u) continue;
    const uint8_t* __restrict__ pix = p.pixels + (size_t)img * p.pixel_stride;
    const uint32_t n = p.npx, lo = s * (64u * K);

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
__global__ __launch_bounds__(64) void enc_scan_groups(EncParams p) {
    const uint32_t lane = lane_id();
    const uint32_t G = blockIdx.x;                       // img * gpi + grp
    const uint32_t img = G / p.gpi, grp = G - img * p.gpi;
    if (p.only_flagged && p.need_generic[img] == 0u) return;
    const uint32_t s0 = grp * 64u;
    const uint32_t s1 = min(p.spi, s0 + 64u);
    uint32_t cur = 0; bool curv = false; int curle = -1;
    for (uint32_t s = s0; s < s1; ++s) {
        const size_t g = (size_t)img * p.spi + s;
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
__global__ __launch_bounds__(64) void enc_scan_images(EncParams p) {
    const uint32_t lane = lane_id();
    const uint32_t img = blockIdx.x;
    if (p.only_flagged && p.need_generic[img] == 0u) return;
    uint32_t cur = 0; int curle = -1;          // table starts zeroed (qoi.h:393), no edge yet
    for (uint32_t gr = 0; gr < p.gpi; ++gr) {
        const size_t G = (size_t)img * p.gpi + gr;
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
// E3: classify + size + look-back + emit, one wavefront per slab.
// ---------------------------------------------------------------------------------

// Literal (non-run, non-index) chunk for px after prev: RGBA / DIFF / LUMA / RGB
// (qoi.h:438-474).  Returns the chunk bytes little-endian in `bytes`, length in `len`.
__device__ __forceinline__ void literal_chunk(uint32_t px, uint32_t prev, u64& bytes, uint32_t& len) {
    const int dr = (int)(int8_t)((px & 0xFF) - (prev & 0xFF));
    const int dg = (int)(int8_t)(((px >> 8) & 0xFF) - ((prev >> 8) & 0xFF));
    const int db = (int)(int8_t)(((px >> 16) & 0xFF) - ((prev >> 16) & 0xFF));
    const int drg = (int)(int8_t)(dr - dg);
    const int dbg = (int)(int8_t)(db - dg);
    const bool alpha_same = ((px ^ prev) >> 24) == 0;
    const bool is_diff = (unsigned)(dr + 2) < 4u && (unsigned)(dg + 2) < 4u && (unsigned)(db + 2) < 4u;
    const bool is_luma = (unsigned)(dg + 32) < 64u && (unsigned)(drg + 8) < 16u && (unsigned)(dbg + 8) < 16u;
    const uint32_t diff_b = kTagDiff | ((dr + 2) << 4) | ((dg + 2) << 2) | (db + 2);
    const uint32_t luma_b = (kTagLuma | (dg + 32)) | ((((drg + 8) << 4) | (dbg + 8)) << 8);
    const u64 rgb_b = (u64)kTagRgb | ((u64)(px & 0x00FFFFFFu) << 8);
    const u64 rgba_b = (u64)kTagRgba | ((u64)px << 8);
    if (!alpha_same) { bytes = rgba_b; len = 5; }
    else if (is_diff) { bytes = diff_b; len = 1; }
    else if (is_luma) { bytes = luma_b; len = 2; }
    else { bytes = rgb_b; len = 4; }
}

// PROBE selects how the colour table is probed/updated for the 64 pixels of a step:
//   0  ds_or_b64 lane masks + ds_bpermute (order-independent, always valid)
//   1  one ds_wrxchg_rtn_b32 per edge lane: relies on the LDS serving the lanes of one
//      instruction that hit the same address in ascending lane order - MEASURED at context
//      creation by lds_order_selftest; the host only picks 1 when that test passes.
// LAST: the slab holds the image's last pixel (and possibly lanes beyond it).
// ABL:  ablation bits for profiling only (1: no emission, 2: no look-back, 4: no probe).
//
// Instruction budget (profiles/r01_s3_issue_rates.txt): a wave64 VALU op costs ~1.25 ns of a SIMD,
// an SALU op ~1.8 ns (one scalar unit per CU) and overlaps with VALU only up to about half the
// VALU count, a DS op 4 LDS cycles of the CU whatever its width.  So the step below is written
// for few instructions of every kind:
//   * every pixel emits at most ONE chunk: a repeat pixel carries the run byte of the run it
//     closes (qoi.h:417-421,425-428 put that byte in front of the next edge's chunk, which is
//     the same stream position), so chunk length is in {0,1,2,4,5} per lane;
//   * per-lane predicates live as 64-bit lane masks in SGPRs (ballot results); the few mask
//     combinations are explicit scalar ops and come back as exec / v_cndmask masks through
//     inverse_ballot;
//   * work that a step does not need is skipped by wave-uniform branches (no repeats: no run
//     arithmetic; no edges: no hash/probe/deltas; no literal: no deltas; no 4/5-byte chunk: no
//     third offset count).
// Chunk bytes go to a per-wave LDS staging buffer at slab-local offsets; once the slab's byte
// offset is known the staged bytes are copied out with aligned dword stores.
template <int K>
struct EncLds {
    static constexpr uint32_t kStageBytes = 64u * K * 5u + 8u;     // <= 5 B/px, + slack
    static constexpr uint32_t kStageDwords = ((kStageBytes + 15u) / 16u) * 4u;
    alignas(256) uint32_t table[64];   // 256-byte aligned: slot address = base | (4*slot)
    u64 mask[64];                      // PROBE 0 only
    alignas(16) uint32_t stage[kStageDwords];
};

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
// no trip through an SGPR.
__device__ __forceinline__ uint32_t prev_pixels(uint32_t v, uint32_t before) {
    const int first = __builtin_amdgcn_mov_dpp((int)before, 0x13C, 0xf, 0xf, false);              // wave_ror:1 (every lane has a source)
    return (uint32_t)__builtin_amdgcn_update_dpp(first, (int)v, 0x138, 0xf, 0xf, false);          // wave_shr:1, lane 0 keeps `first`
}

// Byte 0 of every chunk and byte 1 of the 2+-byte chunks of one step into the LDS staging buffer.
// Runs with all 64 lanes enabled (the slab loop is wave-uniform), so exec is switched with plain
// moves: 3 scalar + 1 vector op around the two stores.  LDS ops of a wave complete in issue order,
// so the later (compiler-visible) reads of the staging buffer see these stores.
// a lane mask the compiler may have lost track of as wave-uniform -> SGPR pair (free when it already is one)
__device__ __forceinline__ u64 uniform64(u64 m) {
    // (the builtin returns int: without the casts to uint32_t a low half with bit 31 set sign-extends over the high half - the
    // generic path then took 32 slots of the image-level table for group-local ones whenever slot 31 had been written in the group)
    return (u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)m) | ((u64)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(m >> 32)) << 32);
}
__device__ __forceinline__ void stage_short(uint32_t addr, uint32_t w, u64 any, u64 second) {
    uint32_t hi;
    asm volatile("s_mov_b64 exec, %3\n\t"
                 "ds_write_b8 %1, %2\n\t"
                 "v_lshrrev_b32 %0, 8, %2\n\t"
                 "s_mov_b64 exec, %4\n\t"
                 "ds_write_b8 %1, %0 offset:1\n\t"
                 "s_mov_b64 exec, -1"
                 : "=&v"(hi) : "v"(addr), "v"(w), "s"(any), "s"(second) : "memory");
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
__device__ __forceinline__ void probe_wait(uint32_t& seen) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(seen) : : "memory"); }

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

// a != b on the scalar unit (hipcc turns a uniform bool -> int into v_cndmask and the masks built from it into VGPRs)
__device__ __forceinline__ uint32_t scalar_ne(uint32_t a, uint32_t b) {
    uint32_t r;
    asm("s_cmp_lg_u32 %1, %2\n\ts_cselect_b32 %0, 1, 0" : "=s"(r) : "s"(a), "s"(b) : "scc");
    return r;
}

// chunk word tags (bits 16..18, above the two bytes a short chunk stores): length class of the lane
constexpr uint32_t kLenOne = 0u, kLenTwo = 1u << 17, kLenLong = 1u << 18;   // 1-byte chunks need no tag: nothing tests for them

// Everything a slab reads from global memory before its first step; fetched one slab ahead so
// that every wavefront always has a slab's worth of loads in flight (without it the kernel is
// bound by HBM latency: ~3 TB/s whatever the content).
template <int K>
struct SlabIn {
    uint32_t warm[8];            // ENTRY 1: the 512 pixels before the slab (step k: pixels lo-64(k+1) .. +63), and the one before them
    uint32_t warm_carry;
    uint32_t px[K];              // pixel t*64 + lane of the slab
    uint32_t carry0;             // pixel before the slab (qoi.h:396-399 start value for slab 0)
    uint32_t next_first;         // first pixel of the next slab
    uint32_t tab_loc, tab_far;   // entry colour table: group-local part / image-level part (lane = slot)
    u64 tab_valid;
    int le_loc, le_far;          // last edge before the slab: group-local / image-level
};

template <int CH, int K, int ENTRY>
__device__ __forceinline__ void load_slab(const EncParams& p, uint32_t g, uint32_t lane, SlabIn<K>& in) {
    const uint32_t img = g / p.spi, s = g - img * p.spi;
    const uint8_t* __restrict__ pix = p.pixels + (size_t)img * p.pixel_stride;
    const uint32_t n = p.npx, lo = s * (64u * K);
    if (ENTRY == 1) {                                        // issued first: they are needed first
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int i = (int)lo - 64 * (k + 1) + (int)lane;
            in.warm[k] = i >= 0 ? load_px<CH>(pix, (uint32_t)i) : kInitPx;
        }
        const int ci = (int)lo - 64 * 8 - 1;
        in.warm_carry = ci >= 0 ? load_px<CH>(pix, (uint32_t)ci) : kInitPx;
    }
    if (lo + 64u * K < n) {                                  // interior slab: no bounds checks
#pragma unroll
        for (int t = 0; t < K; ++t) in.px[t] = load_px<CH>(pix, lo + t * 64u + lane);
        in.next_first = load_px<CH>(pix, lo + 64u * K);
    } else {
#pragma unroll
        for (int t = 0; t < K; ++t) { const uint32_t i = lo + t * 64u + lane; in.px[t] = i < n ? load_px<CH>(pix, i) : 0u; }
        in.next_first = 0;
    }
    in.carry0 = (lo > 0) ? load_px<CH>(pix, lo - 1) : kInitPx;
    if (ENTRY == 0) {
        const uint32_t G = img * p.gpi + (s >> 6);
        in.tab_loc = p.ent_tab[(size_t)g * 64u + lane];
        in.tab_far = p.gent_tab[(size_t)G * 64u + lane];
        in.tab_valid = p.ent_valid[g];
        in.le_loc = p.ent_le[g];
        in.le_far = p.gent_le[G];
    }
}

// ENTRY 1: a slab finds its entry state itself.  The colour table before pixel `lo` is "the last edge pixel
// per hash slot" (qoi.h:430-436), so the wavefront walks BACKWARDS over the pixels before its slab, 64 at a
// time, and fills every slot that is still empty with the latest edge pixel that hashes there, until all 64
// slots are known or the image start is reached (untouched slots are then the zeroes of qoi.h:393).  Natural
// images and noise need 5-8 steps (SURVEY: all 64 slots are rewritten within ~700 pixels); flat content does
// not finish within the window - the slab then flags its image and the generic passes (per-slab summaries +
// scans, ENTRY 0) redo that image.  Unfilled slots hold slot+1, a value that cannot hash to its own slot
// (3(s+1) != s mod 64).  Returns false if the window did not suffice.
constexpr int kWarmSteps = 128;      // look-back window: 8192 pixels (natural content is done after 5-11 steps)
constexpr int kWarmBatch = 8;        // 64-pixel steps loaded together
constexpr int kWarmMinFilled = 48;   // slots that must be known after the first batch (512 pixels), else the content is flat: give up

template <int CH, int K>
__device__ __forceinline__ bool warm_entry_state(const EncParams& p, uint32_t img, uint32_t lo, uint32_t lane,
                                                 EncLds<K>& L, uint32_t tbase, const SlabIn<K>& in, int& last_edge) {
    last_edge = -1;
    if (lo == 0u) { L.table[lane] = 0u; return true; }       // qoi.h:393: zeroed table, no edge yet
    const uint8_t* __restrict__ pix = p.pixels + (size_t)img * p.pixel_stride;
    const uint32_t sent = lane + 1u;
    L.table[lane] = sent;
    __builtin_amdgcn_wave_barrier();
    // ---- the 512 pixels right before the slab, oldest first: later edge pixels simply overwrite earlier ones
    //      (these loads were issued ahead of the slab's own pixels) ------------------------------------------
    {
        const uint32_t carry = __builtin_amdgcn_readfirstlane(in.warm_carry);
#pragma unroll
        for (int k = kWarmBatch - 1; k >= 0; --k) {
            const int base = (int)lo - 64 * (k + 1);
            const uint32_t px = in.warm[k];
            const uint32_t prev = k + 1 < kWarmBatch ? prev_pixels(px, in.warm[k + 1 < kWarmBatch ? k + 1 : k]) : from_lane_below(px, carry);
            const u64 E = __ballot(px != prev);               // lanes before the image start hold the start pixel: no edge
            if (E) {
                last_edge = base + msb64(E);
                (void)probe_swap(tbase | slot_byte_offset(px), px, E);
            }
        }
    }
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
            if (want) (void)probe_swap(tbase | so, px, want);            // same slot twice in a step: the later pixel wins
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

template <int CH, int K, int PROBE, bool LAST, int ABL, int ENTRY>
__device__ __forceinline__ void encode_one_slab(const EncParams& p, uint32_t g, uint32_t lane, EncLds<K>& L, const SlabIn<K>& in) {
    const uint32_t img = g / p.spi, s = g - img * p.spi;
    const uint32_t n = p.npx, lo = s * (64u * K);
    uint8_t* stage8 = reinterpret_cast<uint8_t*>(L.stage);
    const uint32_t* cur = in.px;
    // same value in every lane, but loaded into VGPRs: make them provably wave-uniform
    const uint32_t carry0 = __builtin_amdgcn_readfirstlane(in.carry0);
    const uint32_t next_first = __builtin_amdgcn_readfirstlane(in.next_first);

    // ---- entry state: colour table + distance to the last edge ---------------------------
    int last_edge;
    if (ENTRY == 1) {
        uint32_t tb = lds_addr(L.table);
        asm volatile("" : "+v"(tb));
        if (!warm_entry_state<CH, K>(p, img, lo, lane, L, tb, in, last_edge)) {
            if (lane == 0) { atomicOr(&p.need_generic[img], 1u); atomicOr(p.any_generic, 1u); }
            return;
        }
    } else {
        const u64 lv = uniform64(in.tab_valid);
        L.table[lane] = ((lv >> lane) & 1ull) ? in.tab_loc : in.tab_far;
        last_edge = max(__builtin_amdgcn_readfirstlane(in.le_loc), __builtin_amdgcn_readfirstlane(in.le_far));   // max edge position < lo, or -1
    }
    if (PROBE == 0) L.mask[lane] = 0;
    // ccp = 63 + (first pixel of the step - last edge before the step): stands in for clz(edges below the lane)
    uint32_t ccp = 63u + (uint32_t)((int)lo - last_edge);
    __builtin_amdgcn_wave_barrier();

    // lane constants
    const uint32_t below_lo = lane < 32u ? (1u << lane) - 1u : 0xFFFFFFFFu;
    const uint32_t below_hi = lane < 32u ? 0u : (1u << (lane - 32u)) - 1u;
    const uint32_t lane_run = lane + 128u + kLenOne;       // + clz(edges below) = 0xBF + run length, tagged 1-byte
    uint32_t tbase = lds_addr(L.table);                    // 256-byte aligned
    asm volatile("" : "+v"(tbase));                        // keep in a VGPR: slot address = one v_and_or
    const uint32_t sbase = lds_addr(L.stage);

    uint32_t spos = 0;                                     // bytes staged so far (wave-uniform)
    // edges of step 0
    uint32_t prev_n = from_lane_below(cur[0], carry0);
    u64 E = __ballot(cur[0] != prev_n);
    if (LAST) { const uint32_t r = n - lo; if (r < 64u) E &= (1ull << r) - 1ull; }
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const uint32_t px = cur[t];
        const uint32_t prev = prev_n;
        const u64 Ec = E;
        // ---- lane masks of this step -------------------------------------------------------
        u64 V = ~0ull, lastbit = 0ull;                     // LAST: valid lanes, lane of the image's last pixel
        if (LAST) {
            const int r = (int)(n - lo) - t * 64;         // pixels of the image left at this step
            V = r >= 64 ? ~0ull : (r <= 0 ? 0ull : (1ull << r) - 1ull);
            lastbit = (r >= 1 && r <= 64) ? 1ull << (r - 1) : 0ull;
        }
        // edges of the next step (its lane 0 tells lane 63 whether its run ends here)
        u64 nb;
        if (t + 1 < K) {
            prev_n = prev_pixels(cur[t + 1], px);
            E = __ballot(cur[t + 1] != prev_n);
            if (LAST) { const int r = (int)(n - lo) - (t + 1) * 64; E &= r >= 64 ? ~0ull : (r <= 0 ? 0ull : (1ull << r) - 1ull); }
            nb = E & 1ull;
        } else {
            nb = LAST ? 0ull : (u64)scalar_ne(next_first, read_lane(px, 63));    // 1: the next slab starts with an edge
        }
        if (LAST && V == 0ull) continue;
        const u64 En = (Ec >> 1) | (nb << 63) | lastbit;  // lanes whose successor is an edge (or that end the image)
        const u64 NE = ~Ec & V;                            // repeat pixels
        u64 RB = NE & En;                                  // repeat pixels that close a run: they carry its run byte

        // ---- repeats: run byte 0xC0|(run-1) on the pixel that closes a run (qoi.h:416-421,425-428) ----
        // clz of the edges below the lane; ccp stands in when the run began before this step
        uint32_t w;
        {
            const uint32_t fhi = ffbh((uint32_t)(Ec >> 32) & below_hi);
            const uint32_t flo = ffbh((uint32_t)Ec & below_lo) | 32u;
            const uint32_t m = min(min(fhi, flo), ccp);
            w = m + lane_run;                              // 0xBF + count (tagged), count = repeats since the last edge
            if (__ballot(w > (0xFCu | kLenOne)) & NE) {    // some run reaches 62: wave-uniform slow path (flat content)
                const uint32_t cnt = w - (0xBFu | kLenOne);
                const uint32_t xm = cnt % 62u;
                w = (xm ? 0xBFu + xm : 0xFDu) | kLenOne;   // a repeat landing on a multiple of 62 closes a full run
                RB |= NE & __ballot(xm == 0u);
            }
        }
        const u64 any = Ec | RB;                           // lanes that emit a chunk
        if (Ec) {
            ccp = (uint32_t)__builtin_clzll(Ec) + 64u;
            // ---- colour-table probe/update (qoi.h:430-436) for edge pixels ---------------------
            const uint32_t hsh = __builtin_amdgcn_udot4(px, 0x2C1C140Cu, 0u, false);   // 4 * QOI_COLOR_HASH (qoi.h:322)
            uint32_t seen = ~px;
            if (!(ABL & 4)) {
                if (PROBE == 1) {
                    seen = probe_swap((hsh & 0xFCu) | tbase, px, Ec);
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
            }
            // ---- chunk of an edge pixel (qoi.h:432-474): INDEX, else RGBA if alpha moved, else DIFF, LUMA, RGB ----
            // wrapped byte deltas live in the low byte of d*; consumers sign-extend that byte (SDWA)
            const uint32_t d_r = px - prev, d_g = sub_byte1(px, prev), d_b = sub_byte2(px, prev);
            const uint32_t tr = (int)(int8_t)d_r + 2, tg = (int)(int8_t)d_g + 2, tb = (int)(int8_t)d_b + 2;
            const uint32_t tg8 = (int)(int8_t)d_g - 6, ug = (int)(int8_t)d_g + 32;
            const uint32_t ur = tr - tg8, ub = tb - tg8;   // dr-dg+8, db-dg+8
            const bool is_diff = (tr | tg | tb) < 4u;
            const bool is_luma = ((ug >> 2) | ur | ub) < 16u;
            const bool is_ad = (px ^ prev) > 0x00FFFFFFu;  // alpha differs
            const uint32_t w_diff = (kTagDiff | kLenOne) | (tr << 4) | (tg << 2) | tb;
            const uint32_t w_luma = (kTagLuma | kLenTwo | ug) | (ur << 12) | (ub << 8);
            uint32_t we = is_luma ? w_luma : kLenLong;
            we = is_diff ? w_diff : we;
            we = is_ad ? kLenLong : we;
            asm volatile("" : "+v"(we));                   // keep the literal classes branch-free (no sinking under !hit)
            if (PROBE == 1 && !(ABL & 4)) probe_wait(seen);
            const bool is_hit = seen == px;
            we = is_hit ? (((hsh >> 2) & 63u) | kLenOne) : we;         // QOI_OP_INDEX (qoi.h:432-434)
            w = in_mask(Ec) ? we : w;
            const u64 lng = __ballot(w >= kLenLong);
            if (__builtin_expect(lng != 0ull, 0)) {
                // rare in natural images: some lane carries QOI_OP_RGB / QOI_OP_RGBA (qoi.h:461-474)
                const u64 five = lng & __ballot(is_ad);
                const u64 two = __ballot(w >= kLenTwo) & ~lng;
                const u64 b0 = (any & ~(two | lng)) | five;            // odd lengths: 1-byte chunks and RGBA
                const uint32_t w_long = (px << 8) | (in_mask(five) ? kTagRgba : kTagRgb);   // tag r g b (a follows for RGBA)
                w = in_mask(lng) ? w_long : w;
                const uint32_t off = sbase + spos + count_below(b0) + 2u * count_below(two) + 4u * count_below(lng);
                spos += (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(two) + 4u * (uint32_t)__builtin_popcountll(lng);
                if (!(ABL & 1)) {
                    stage_short(off, w, any, two | lng);
                    lds_u8* dst = (lds_u8*)off;
                    if (in_mask(lng)) { dst[2] = (uint8_t)(w >> 16); dst[3] = (uint8_t)(w >> 24); }
                    if (in_mask(five)) dst[4] = (uint8_t)(px >> 24);
                }
                continue;
            }
        } else {
            ccp += 64u;
        }
        // ---- common case: chunk lengths 1 and 2 only.  offset = #chunks below + #LUMA chunks below ----
        const u64 two = __ballot(w >= kLenTwo) & any;
        const uint32_t off = count_below_from(two, count_below_from(any, sbase + spos));
        spos += (uint32_t)__builtin_popcountll(any) + (uint32_t)__builtin_popcountll(two);
        if (!(ABL & 1)) stage_short(off, w, any, two);
    }
    const uint32_t slab_pos = spos;

    const uint32_t slab_bytes = slab_pos;
    if (p.scratch) {
        // ---- order-free mode: park the slab's bytes in its scratch slot, E4 compacts ------------
        if (lane == 0) p.slab_size[g] = slab_bytes;
        __builtin_amdgcn_wave_barrier();
        uint4* dst = reinterpret_cast<uint4*>(p.scratch + (size_t)g * kEncScratchStride);
        const uint4* src = reinterpret_cast<const uint4*>(L.stage);
        const uint32_t n16 = (slab_bytes + 15u) >> 4;
        for (uint32_t j = lane; j < n16; j += 64u) dst[j] = src[j];
        return;
    }

    // ---- slab byte count -> offset: decoupled look-back over earlier slabs -----------------
    u64 excl = 0;
    if (!(ABL & 2)) {
        constexpr u64 kAgg = 1ull << 62, kIncl = 2ull << 62, kVal = (1ull << 62) - 1ull;
        u64* st = p.status;
        if (s == 0) {
            if (lane == 0) granule_store(&st[g], kIncl | slab_bytes);
        } else {
            if (lane == 0) granule_store(&st[g], kAgg | slab_bytes);
            const uint32_t first = g - s;                 // global id of this image's slab 0
            int64_t look = (int64_t)g - 1;                // newest slab of the current window
            uint32_t spins = 0;
            bool done = false;
            while (!done) {
                const int64_t mine = look - (int64_t)lane;
                const bool inwin = mine >= (int64_t)first;
                u64 v = inwin ? granule_load(&st[mine]) : kIncl;   // before slab 0: inclusive prefix 0
                const u64 notready = __ballot((v >> 62) == 0);
                const u64 incl = __ballot((v >> 62) == 2);
                const int stop = incl ? __builtin_ctzll(incl) : 64;       // nearest inclusive record
                const u64 need = stop >= 64 ? ~0ull : ((1ull << stop) - 1ull);
                if (notready & need) {                                    // a record we must add is not published yet
                    // (first pass: a predecessor that gave the image up never publishes - the image is encoded again anyway)
                    if (ENTRY == 1 && __hip_atomic_load((gu32*)&p.need_generic[img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) return;
                    if (++spins > (1u << 22)) { if (lane == 0) atomicOr(p.err, 1u); break; }
                    __builtin_amdgcn_s_sleep(1);
                    continue;
                }
                const u64 part = wave_sum64_upto(v & kVal, lane, stop);
                excl += part;
                if (incl) done = true; else look -= 64;
            }
            if (lane == 0) granule_store(&st[g], kIncl | (excl + slab_bytes));
        }
    }

    // ---- copy the staged bytes out ---------------------------------------------------------
    uint8_t* __restrict__ out = p.out + (size_t)img * p.out_stride;
    if (s == 0 && lane < (uint32_t)kHeaderBytes) {        // 14-byte header (qoi.h:384-388)
        const uint32_t w = p.width, h = p.height;
        const u64 hdr_lo = 0x66696F71ull | ((u64)__builtin_bswap32(w) << 32);             // "qoif", width BE
        const u64 hdr_hi = (u64)__builtin_bswap32(h) | ((u64)p.channels << 32) | ((u64)p.colorspace << 40);
        out[lane] = (uint8_t)((lane < 8u ? hdr_lo : hdr_hi) >> (8u * (lane & 7u)));
    }
    const u64 pos = (u64)kHeaderBytes + excl;
    if (!(ABL & 1) && slab_bytes) {
        __builtin_amdgcn_wave_barrier();
        // head up to the first 16-byte boundary byte by byte, aligned 16-byte stores (source re-aligned with v_alignbyte: the
        // staged bytes sit `head` past a dword boundary), tail byte by byte - as enc_compact does from the scratch slot
        uint8_t* dst = out + pos;
        const uint32_t mis = (uint32_t)(uintptr_t)dst & 15u;
        const uint32_t head = min(slab_bytes, (16u - mis) & 15u);
        if (lane < head) dst[lane] = stage8[lane];
        const uint32_t n16 = (slab_bytes - head) >> 4;
        uint4* d16 = reinterpret_cast<uint4*>(dst + head);
        const uint32_t sh = head & 3u;
        for (uint32_t j = lane; j < n16; j += 64u) {
            const uint32_t* q = &L.stage[(head >> 2) + 4u * j];
            const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
            uint4 v;
            v.x = __builtin_amdgcn_alignbyte(w1, w0, sh); v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
            v.z = __builtin_amdgcn_alignbyte(w3, w2, sh); v.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
            d16[j] = v;
        }
        const uint32_t done_b = head + (n16 << 4);
        if (lane < slab_bytes - done_b) dst[done_b + lane] = stage8[done_b + lane];
    }
    if (LAST) {                                           // trailer (qoi.h:339,480-482) + *out_len
        const u64 end = pos + slab_bytes;
        if (lane < (uint32_t)kTrailerBytes) out[end + lane] = (lane == 7u) ? 1 : 0;
        if (lane == 0) p.out_len[img] = (int)(end + kTrailerBytes);
    }
}

// ENTRY 0: entry state from the per-slab summaries + scans (E1/E2).  ENTRY 1: each slab finds it itself
// (warm_entry_state); slabs whose look-back window does not suffice flag their image, and the launcher runs
// the ENTRY 0 passes with only_flagged set: small grid-stride grids that return at once when nothing was
// flagged.  A workgroup serves unit u = (image u % n_images, group u / n_images) so that the slabs in flight
// spread over all images.
template <int CH, int K, int PROBE, int ABL, int ENTRY>
__global__ __launch_bounds__(256, 6) void enc_slabs(EncParams p) {
    __shared__ EncLds<K> s_lds[4];
    __shared__ uint32_t s_ticket;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    if (p.only_flagged && *p.any_generic == 0u) return;
#pragma unroll 1
    for (uint32_t unit = blockIdx.x; unit < p.n_units; unit += gridDim.x) {
        const uint32_t img = unit % p.n_images;
        uint32_t quad = unit / p.n_images;                 // order-free (scratch) mode: any order will do
        if (p.only_flagged && p.need_generic[img] == 0u) continue;
        if (p.use_ticket && !p.scratch) {
            // look-back mode (one unit per workgroup): groups are handed out by the image's ticket counter, i.e. in
            // START order, hence every predecessor a look-back can wait on is already running or finished (no
            // reliance on dispatch order; guide G16).  One counter per image keeps the atomics off a single hot word.
            if (threadIdx.x == 0) s_ticket = atomicAdd(&p.ticket[img], 1u);
            __syncthreads();
            quad = __builtin_amdgcn_readfirstlane(s_ticket);
        }
        uint32_t s_pos = quad * p.quads_per_wg * 4u + wave;
#pragma unroll 1
        for (uint32_t r = 0; r < p.quads_per_wg && s_pos < p.spi; ++r, s_pos += 4u) {
            const uint32_t g = img * p.spi + s_pos;
            if (ENTRY == 1 && __hip_atomic_load((gu32*)&p.need_generic[img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) break;   // image already sent to the generic path
            SlabIn<K> in;
            load_slab<CH, K, ENTRY>(p, g, lane, in);
            if (s_pos == p.spi - 1u) encode_one_slab<CH, K, PROBE, true, ABL, ENTRY>(p, g, lane, s_lds[wave], in);
            else encode_one_slab<CH, K, PROBE, false, ABL, ENTRY>(p, g, lane, s_lds[wave], in);
            __builtin_amdgcn_wave_barrier();
        }
    }
}

// ---------------------------------------------------------------------------------
// E4a: exclusive scan of the slab byte counts of one image (one workgroup per image);
// also writes the 14-byte header (qoi.h:384-388), the 8-byte end marker (qoi.h:339,480-482)
// and *out_len (qoi.h:484).
// ---------------------------------------------------------------------------------
// The scan walks the image in tiles of 16384 slabs, 1024 per wavefront: counts are loaded coalesced (next tile's
// while this one is scanned), turned through a wavefront-private LDS stripe so that a lane holds 16 consecutive
// counts, scanned (lane-serial, then six rounds over the wavefront, then over the 16 wavefronts) and written back
// the same way.  A 4K frame is one tile, a 16384 x 16384 image 16 - the first version gave every thread a
// contiguous 1/256 of the image to add up serially and took 0.4 ms on that image.
__global__ __launch_bounds__(1024) void enc_offsets(EncParams p) {
    constexpr uint32_t kPer = 16, kStripe = 64u * kPer, kTile = 16u * kStripe;
    __shared__ uint32_t s_turn[16][kStripe + 64u];             // element e of a stripe at e + e/16 (bank spread)
    __shared__ uint32_t s_wave[16];
    const uint32_t img = blockIdx.x, tid = threadIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const uint32_t* __restrict__ sz = p.slab_size + (size_t)img * p.spi;
    uint32_t* __restrict__ off = p.slab_off + (size_t)img * p.spi;
    const uint32_t n = p.spi;
    uint32_t* turn = s_turn[wave];
    uint32_t carry = 0;
    uint32_t nv[kPer];
#pragma unroll
    for (uint32_t j = 0; j < kPer; ++j) { const uint32_t e = wave * kStripe + j * 64u + lane; nv[j] = e < n ? sz[e] : 0u; }
    for (uint32_t base = 0; base < n; base += kTile) {
        const uint32_t sbase = base + wave * kStripe;          // first slab of this wavefront's stripe
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
    uint8_t* out = p.out + (size_t)img * p.out_stride;
    const uint32_t total = carry;
    if (tid < (uint32_t)kHeaderBytes) {
        const uint32_t w = p.width, h = p.height;
        const u64 hdr_lo = 0x66696F71ull | ((u64)__builtin_bswap32(w) << 32);             // "qoif", width BE
        const u64 hdr_hi = (u64)__builtin_bswap32(h) | ((u64)p.channels << 32) | ((u64)p.colorspace << 40);
        out[tid] = (uint8_t)((tid < 8u ? hdr_lo : hdr_hi) >> (8u * (tid & 7u)));
    }
    if (tid < (uint32_t)kTrailerBytes) out[(size_t)kHeaderBytes + total + tid] = (tid == 7u) ? 1 : 0;
    if (tid == 0) p.out_len[img] = (int)(kHeaderBytes + total + kTrailerBytes);
}

// E4b: move every slab's bytes from its scratch slot to its place in the stream
// (one wavefront per slab; aligned 16-byte stores, source re-aligned with v_alignbyte).
__global__ __launch_bounds__(256) void enc_compact(EncParams p) {
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t g = blockIdx.x * 4u + wave;
    if (g >= p.n_images * p.spi) return;
    const uint32_t img = g / p.spi;
    const uint32_t n = p.slab_size[g];
    if (n == 0) return;
    const uint8_t* __restrict__ src = p.scratch + (size_t)g * kEncScratchStride;      // 16-byte aligned
    uint8_t* __restrict__ dst = p.out + (size_t)img * p.out_stride + kHeaderBytes + p.slab_off[g];
    const uint32_t mis = (uint32_t)(uintptr_t)dst & 15u;
    const uint32_t head = min(n, (16u - mis) & 15u);                 // bytes before dst becomes 16-byte aligned
    if (lane < head) dst[lane] = src[lane];
    const uint32_t n16 = (n - head) >> 4;
    uint4* __restrict__ d16 = reinterpret_cast<uint4*>(dst + head);
    const uint32_t* __restrict__ s32 = reinterpret_cast<const uint32_t*>(src) + (head >> 2);
    const uint32_t sh = head & 3u;
    for (uint32_t j = lane; j < n16; j += 64u) {
        const uint32_t* q = s32 + 4u * j;
        const uint32_t w0 = q[0], w1 = q[1], w2 = q[2], w3 = q[3], w4 = q[4];
        uint4 v;
        v.x = __builtin_amdgcn_alignbyte(w1, w0, sh); v.y = __builtin_amdgcn_alignbyte(w2, w1, sh);
        v.z = __builtin_amdgcn_alignbyte(w3, w2, sh); v.w = __builtin_amdgcn_alignbyte(w4, w3, sh);
        d16[j] = v;
    }
    const uint32_t done = head + (n16 << 4);
    if (lane < n - done) dst[done + lane] = src[done + lane];
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
template <int CH, int K, int PROBE, int ABL>
static void launch_encode_t(EncParams p, hipStream_t st, KernelTimer* tm, int phases) {
    const uint32_t total = p.n_images * p.spi;
    const uint32_t blocks = (total + 3u) / 4u;
    const uint32_t quads_per_image = (p.spi + 3u) / 4u;
    const uint32_t wgs_per_image = (quads_per_image + p.quads_per_wg - 1u) / p.quads_per_wg;
    p.n_units = wgs_per_image * p.n_images;
    const bool warm = p.warm && PROBE == 1;
    uint32_t small = 2048u;                                  // grid of the passes that usually have nothing to do
    tm->mark(kT_begin, st);
    if (phases & kEncSlabs) {
    if (warm) {
        p.only_flagged = 0;
        hipLaunchKernelGGL((enc_slabs<CH, K, PROBE, ABL, 1>), dim3(p.n_units), dim3(256), 0, st, p);
        tm->mark(kT_enc_slabs, st);
        p.only_flagged = 1;
    } else {
        p.only_flagged = 0;
        small = 0xFFFFFFFFu;
    }
    hipLaunchKernelGGL((enc_slab_summary<CH, K>), dim3(blocks < small ? blocks : small), dim3(256), 0, st, p);
    tm->mark(kT_enc_summary, st);
    hipLaunchKernelGGL(enc_scan_groups, dim3(p.n_images * p.gpi), dim3(64), 0, st, p);
    tm->mark(kT_enc_scan_groups, st);
    hipLaunchKernelGGL(enc_scan_images, dim3(p.n_images), dim3(64), 0, st, p);
    tm->mark(kT_enc_scan_images, st);
    // look-back mode: the images the first pass gave up on are encoded again from their first slab, with look-back records and
    // tickets of their own (those of the first pass are spent)
    if (warm && !p.scratch) { p.status = p.status2; p.ticket = p.ticket2; }
    hipLaunchKernelGGL((enc_slabs<CH, K, PROBE, ABL, 0>), dim3(p.n_units < small ? p.n_units : small), dim3(256), 0, st, p);
    tm->mark(warm ? kT_enc_slabs_generic : kT_enc_slabs, st);
    }
    if (p.scratch && (phases & kEncPlace)) {
        hipLaunchKernelGGL(enc_offsets, dim3(p.n_images), dim3(1024), 0, st, p);
        tm->mark(kT_enc_offsets, st);
        hipLaunchKernelGGL(enc_compact, dim3(blocks), dim3(256), 0, st, p);
        tm->mark(kT_enc_compact, st);
    }
}

void launch_encode(const EncParams& p, hipStream_t st, KernelTimer* tm, int phases) {
    if (p.channels == 3) {
        if (p.probe_xchg) launch_encode_t<3, kEncSteps, 1, 0>(p, st, tm, phases); else launch_encode_t<3, kEncSteps, 0, 0>(p, st, tm, phases);
        return;
    }
    if (!p.probe_xchg) { launch_encode_t<4, kEncSteps, 0, 0>(p, st, tm, phases); return; }
    launch_encode_t<4, kEncSteps, 1, 0>(p, st, tm, phases);
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
