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
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    const uint32_t g = blockIdx.x * 4u + wave;
    const uint32_t total = p.n_images * p.spi;
    if (g >= total) return;
    const uint32_t img = g / p.spi, s = g - img * p.spi;
    const uint8_t* __restrict__ pix = p.pixels + (size_t)img * p.pixel_stride;
    const uint32_t n = p.npx, lo = s * (64u * K);

    s_key[wave][lane] = 0;
    uint32_t carry = (lo > 0) ? load_px<CH>(pix, lo - 1) : kInitPx;
    int le = -1;
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (int t = 0; t < K; ++t) {
        const uint32_t i = lo + t * 64u + lane;
        const bool inb = i < n;
        const uint32_t px = inb ? load_px<CH>(pix, i) : 0u;
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
}

// ---------------------------------------------------------------------------------
// E2a: exclusive "latest valid per slot" / max scan over the <=64 slabs of one group.
// lane = hash slot.  Writes per-slab group-local entry state and the group aggregate.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void enc_scan_groups(EncParams p) {
    const uint32_t lane = lane_id();
    const uint32_t G = blockIdx.x;                       // img * gpi + grp
    const uint32_t img = G / p.gpi, grp = G - img * p.gpi;
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
// Chunk bytes go to a per-wave LDS staging buffer at slab-local offsets while the slab is
// classified (one pass over the pixels); once the look-back has produced the slab's byte
// offset the staged bytes are copied out with aligned dword stores.
// Per-lane predicates are kept as booleans so that hipcc holds them as 64-bit lane masks
// in SGPRs: class algebra and length bits cost scalar instructions, not VALU.
template <int K>
struct EncLds {
    static constexpr uint32_t kStageBytes = 64u * K * 5u + 8u;     // <= 5 B/px + one flushed run byte, + slack
    static constexpr uint32_t kStageDwords = ((kStageBytes + 15u) / 16u) * 4u;
    alignas(16) uint32_t stage[kStageDwords];
    uint32_t table[64];
    uint32_t dummy[64];          // exchange target of lanes that must not touch the table (PROBE 1); must follow table
    u64 mask[64];
};

constexpr int kEncUnroll = 8;     // steps per unrolled group (pixels of the next group are prefetched)

template <int CH, int K, int PROBE, bool LAST, int ABL>
__device__ __forceinline__ void encode_one_slab(const EncParams& p, uint32_t g, uint32_t lane, EncLds<K>& L) {
    const uint32_t img = g / p.spi, s = g - img * p.spi;
    const uint8_t* __restrict__ pix = p.pixels + (size_t)img * p.pixel_stride;
    const uint32_t n = p.npx, lo = s * (64u * K);
    uint8_t* stage8 = reinterpret_cast<uint8_t*>(L.stage);
    uint8_t* table8 = reinterpret_cast<uint8_t*>(L.table);

    // ---- entry state: colour table + last edge position ------------------------------
    {
        const uint32_t G = img * p.gpi + (s >> 6);
        const uint32_t loc = p.ent_tab[(size_t)g * 64u + lane];
        const uint32_t far = p.gent_tab[(size_t)G * 64u + lane];
        const u64 lv = p.ent_valid[g];
        L.table[lane] = ((lv >> lane) & 1ull) ? loc : far;
        if (PROBE == 0) L.mask[lane] = 0;
    }
    int last_edge = max(p.ent_le[g], p.gent_le[img * p.gpi + (s >> 6)]);   // max edge position < lo, or -1
    uint32_t carry = (lo > 0) ? load_px<CH>(pix, lo - 1) : kInitPx;
    __builtin_amdgcn_wave_barrier();

    const u64 lane_bit = 1ull << lane;
    const u64 below = lane_bit - 1ull;

    uint32_t cur[kEncUnroll], nxt[kEncUnroll];
#pragma unroll
    for (int u = 0; u < kEncUnroll; ++u) {
        const uint32_t i = lo + u * 64u + lane;
        cur[u] = (!LAST || i < n) ? load_px<CH>(pix, i) : 0u;
    }
    uint32_t slab_pos = 0;                                 // bytes staged so far (wave-uniform)
#pragma unroll 1
    for (int t0 = 0; t0 < K; t0 += kEncUnroll) {
        if (t0 + kEncUnroll < K) {                          // prefetch the next group's pixels
#pragma unroll
            for (int u = 0; u < kEncUnroll; ++u) {
                const uint32_t i = lo + (uint32_t)(t0 + kEncUnroll + u) * 64u + lane;
                nxt[u] = (!LAST || i < n) ? load_px<CH>(pix, i) : 0u;
            }
        }
#pragma unroll
        for (int u = 0; u < kEncUnroll; ++u) {
            const uint32_t base = lo + (uint32_t)(t0 + u) * 64u;
            const uint32_t i = base + lane;
            const uint32_t px = cur[u];
            const uint32_t prev = from_lane_below(px, carry);
            carry = read_lane(px, 63);
            const bool inb = !LAST || i < n;
            const bool edge = LAST ? (inb && px != prev) : (px != prev);
            const u64 E = __ballot(edge);

            // d = distance to the last edge strictly before this pixel (>= 1)
            const u64 eb = E & below;
            const uint32_t d = eb ? lane - (uint32_t)msb64(eb) : lane + (uint32_t)((int)base - last_edge);
            if (E) last_edge = (int)base + msb64(E);
            const uint32_t x = d - (edge ? 1u : 0u);
            uint32_t xm = x;
            if (__ballot(x >= 62u)) {                     // wave-uniform: long runs are rare in busy content
                xm = x % 62u;
                asm volatile("" : "+v"(xm));              // keep this a real branch (the divide is quarter-rate)
            }
            // an edge flushes pending repeats (qoi.h:425-428); a repeat flushes at 62 or at the last pixel (qoi.h:417)
            bool er = edge ? (xm != 0u) : (xm == 0u);
            if (LAST) er = inb && (er || (!edge && i == n - 1u));
            const uint32_t runb = xm ? 0xBFu + xm : 0xFDu;    // 0xC0|(xm-1); a repeat landing on xm == 0 closes a run of 62

            // ---- colour-table probe/update (qoi.h:430-436) for edge pixels -----------------
            const uint32_t so = slot_byte_offset(px);
            uint32_t seen = ~px;
            if (!(ABL & 4)) {
                if (PROBE == 1) {
                    // every lane exchanges (no exec-masked block -> the scheduler can overlap the LDS round trip
                    // with the delta arithmetic below); repeats hit a private dummy word instead of the table
                    const uint32_t addr = edge ? so : 256u + 4u * lane;
                    seen = __hip_atomic_exchange(reinterpret_cast<uint32_t*>(table8 + addr), px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                } else {
                    if (edge) __hip_atomic_fetch_or(&L.mask[so >> 2], lane_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    __builtin_amdgcn_wave_barrier();
                    const u64 same = edge ? L.mask[so >> 2] : 0ull;    // edge lanes of this step sharing the slot
                    const uint32_t tval = L.table[so >> 2];
                    __builtin_amdgcn_wave_barrier();
                    if (edge) L.mask[so >> 2] = 0;
                    const u64 pred = same & below;
                    const uint32_t pv = gather_lane(px, pred ? (uint32_t)msb64(pred) : lane);
                    seen = pred ? pv : tval;                           // nearest earlier same-slot edge, else carried table
                    if (edge && ((same >> lane) >> 1) == 0) L.table[so >> 2] = px;   // last lane per slot updates the table
                    __builtin_amdgcn_wave_barrier();
                }
            }
            const bool hit = edge && seen == px;

            // ---- classes (qoi.h:432-474 priority: INDEX, then RGBA if alpha moved, DIFF, LUMA, RGB) ----
            const int dr = (int)(int8_t)((px & 0xFF) - (prev & 0xFF));
            const int dg = (int)(int8_t)(((px >> 8) & 0xFF) - ((prev >> 8) & 0xFF));
            const int db = (int)(int8_t)(((px >> 16) & 0xFF) - ((prev >> 16) & 0xFF));
            const int drg = (int)(int8_t)(dr - dg);
            const int dbg = (int)(int8_t)(db - dg);
            const bool lit = edge && !hit;
            const bool c_rgba = lit && ((px ^ prev) >> 24) != 0;
            const bool t_diff = ((unsigned)(dr + 2) | (unsigned)(dg + 2) | (unsigned)(db + 2)) < 4u;
            const bool t_luma = (((unsigned)(dg + 32) >> 2) | (unsigned)(drg + 8) | (unsigned)(dbg + 8)) < 16u;
            const bool c_diff = lit && !c_rgba && t_diff;
            const bool c_luma = lit && !c_rgba && !t_diff && t_luma;
            const bool c_rgb = lit && !c_rgba && !t_diff && !t_luma;
            const bool c_one = hit || c_diff;             // 1-byte chunks
            // length = ll + er with ll in {1 (c_one), 2 (c_luma), 4 (c_rgb), 5 (c_rgba)}: bit algebra on lane masks
            const bool odd = c_one || c_rgba;
            const bool l0 = odd != er, l1 = c_luma || (odd && er), l2 = c_rgb || c_rgba;
            const u64 b0 = __ballot(l0), b1 = __ballot(l1), b2 = __ballot(l2);

            if (!(ABL & 1)) {
                // first byte / second byte of the literal chunk
                const uint32_t diff_b = kTagDiff | ((dr + 2) << 4) | ((dg + 2) << 2) | (db + 2);
                const uint32_t luma_b = (kTagLuma | (dg + 32)) | ((((drg + 8) << 4) | (dbg + 8)) << 8);
                uint32_t w = (px << 8) | (c_rgba ? kTagRgba : kTagRgb);   // tag r g b   (a follows for RGBA)
                w = c_luma ? luma_b : w;
                w = c_diff ? diff_b : w;
                w = hit ? (so >> 2) : w;
                const uint32_t off = slab_pos + count_below(b0) + 2u * count_below(b1) + 4u * count_below(b2);
                uint8_t* dst = stage8 + off;
                if (er) dst[0] = (uint8_t)runb;
                dst += er ? 1 : 0;
                if (edge) dst[0] = (uint8_t)w;
                if (__ballot(c_luma || l2)) {             // some lane has a multi-byte literal
                    if (c_luma || l2) dst[1] = (uint8_t)(w >> 8);
                    if (b2) {
                        if (l2) { dst[2] = (uint8_t)(w >> 16); dst[3] = (uint8_t)(w >> 24); }
                        if (c_rgba) dst[4] = (uint8_t)(px >> 24);
                    }
                }
            }
            slab_pos += (uint32_t)__builtin_popcountll(b0) + 2u * (uint32_t)__builtin_popcountll(b1) + 4u * (uint32_t)__builtin_popcountll(b2);
        }
#pragma unroll
        for (int u = 0; u < kEncUnroll; ++u) cur[u] = nxt[u];
    }

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
        uint8_t* dst = out + pos;
        const uint32_t mis = (uint32_t)(uintptr_t)dst & 3u;
        const uint32_t head = min(slab_bytes, (4u - mis) & 3u);           // bytes up to the first aligned dword
        if (lane < head) dst[lane] = stage8[lane];
        const uint32_t ndw = (slab_bytes - head) >> 2;
        uint32_t* dst32 = reinterpret_cast<uint32_t*>(dst + head);
        for (uint32_t j = lane; j < ndw; j += 64u) {                      // staged bytes sit `head` past a dword boundary
            const uint32_t w0 = L.stage[j], w1 = L.stage[j + 1u];
            dst32[j] = __builtin_amdgcn_alignbyte(w1, w0, head);
        }
        const uint32_t done_b = head + (ndw << 2);
        if (lane < slab_bytes - done_b) dst[done_b + lane] = stage8[done_b + lane];
    }
    if (LAST) {                                           // trailer (qoi.h:339,480-482) + *out_len
        const u64 end = pos + slab_bytes;
        if (lane < (uint32_t)kTrailerBytes) out[end + lane] = (lane == 7u) ? 1 : 0;
        if (lane == 0) p.out_len[img] = (int)(end + kTrailerBytes);
    }
}

template <int CH, int K, int PROBE, int ABL>
__global__ __launch_bounds__(256) void enc_slabs(EncParams p) {
    __shared__ EncLds<K> s_lds[4];
    __shared__ uint32_t s_ticket;
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    // A workgroup serves image (blockIdx % n_images): the slabs in flight spread over all images, so
    // every per-image look-back chain has few unfinished predecessors.  Within its image the
    // workgroup takes the next `quads_per_wg` groups of 4 slabs from the image's ticket counter:
    // slab ids are handed out in START order, hence every predecessor a look-back can wait on
    // is already running or finished (no reliance on dispatch order; guide G16).  One counter
    // per image keeps the atomics off a single hot word.
    const uint32_t img = blockIdx.x % p.n_images;
    uint32_t quad = blockIdx.x / p.n_images;              // order-free (scratch) mode: any order will do
    if (p.use_ticket && !p.scratch) {
        if (threadIdx.x == 0) s_ticket = atomicAdd(&p.ticket[img], 1u);
        __syncthreads();
        quad = s_ticket;
    }
    const uint32_t first_quad = quad * p.quads_per_wg;
#pragma unroll 1
    for (uint32_t r = 0; r < p.quads_per_wg; ++r) {
        const uint32_t s_pos = (first_quad + r) * 4u + wave;
        if (s_pos >= p.spi) return;
        const uint32_t g = img * p.spi + s_pos;
        if (s_pos == p.spi - 1u) encode_one_slab<CH, K, PROBE, true, ABL>(p, g, lane, s_lds[wave]);
        else encode_one_slab<CH, K, PROBE, false, ABL>(p, g, lane, s_lds[wave]);
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------
// E4a: exclusive scan of the slab byte counts of one image (one workgroup per image);
// also writes the 14-byte header (qoi.h:384-388), the 8-byte end marker (qoi.h:339,480-482)
// and *out_len (qoi.h:484).
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc_offsets(EncParams p) {
    __shared__ uint32_t s_part[256];
    const uint32_t img = blockIdx.x, tid = threadIdx.x;
    const uint32_t* __restrict__ sz = p.slab_size + (size_t)img * p.spi;
    uint32_t* __restrict__ off = p.slab_off + (size_t)img * p.spi;
    const uint32_t per = (p.spi + 255u) / 256u;
    const uint32_t lo = tid * per, hi = min(p.spi, lo + per);
    uint32_t sum = 0;
    for (uint32_t i = lo; i < hi; ++i) sum += sz[i];
    s_part[tid] = sum;
    __syncthreads();
    for (uint32_t d = 1; d < 256u; d <<= 1) {           // Hillis-Steele inclusive scan of the partials
        const uint32_t v = tid >= d ? s_part[tid - d] : 0u;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    uint32_t run = tid ? s_part[tid - 1] : 0u;
    for (uint32_t i = lo; i < hi; ++i) { off[i] = run; run += sz[i]; }
    uint8_t* out = p.out + (size_t)img * p.out_stride;
    const uint32_t total = s_part[255];
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
    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
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
__global__ __launch_bounds__(64) void lds_order_selftest(uint32_t* out) {
    __shared__ uint32_t tab[64];
    const uint32_t lane = lane_id();
    uint32_t bad = 0;
    uint32_t rng = 0x9E3779B9u * (blockIdx.x + 1u) + lane * 0x85EBCA6Bu;
    for (int it = 0; it < 256; ++it) {
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
static void launch_encode_t(const EncParams& p, hipStream_t st, KernelTimer* tm) {
    const uint32_t total = p.n_images * p.spi;
    const uint32_t blocks = (total + 3u) / 4u;
    tm->mark(kT_begin, st);
    hipLaunchKernelGGL((enc_slab_summary<CH, K>), dim3(blocks), dim3(256), 0, st, p);
    tm->mark(kT_enc_summary, st);
    hipLaunchKernelGGL(enc_scan_groups, dim3(p.n_images * p.gpi), dim3(64), 0, st, p);
    tm->mark(kT_enc_scan_groups, st);
    hipLaunchKernelGGL(enc_scan_images, dim3(p.n_images), dim3(64), 0, st, p);
    tm->mark(kT_enc_scan_images, st);
    const uint32_t quads_per_image = (p.spi + 3u) / 4u;
    const uint32_t wgs_per_image = (quads_per_image + p.quads_per_wg - 1u) / p.quads_per_wg;
    hipLaunchKernelGGL((enc_slabs<CH, K, PROBE, ABL>), dim3(wgs_per_image * p.n_images), dim3(256), 0, st, p);
    tm->mark(kT_enc_slabs, st);
    if (p.scratch) {
        hipLaunchKernelGGL(enc_offsets, dim3(p.n_images), dim3(256), 0, st, p);
        tm->mark(kT_enc_offsets, st);
        hipLaunchKernelGGL(enc_compact, dim3(blocks), dim3(256), 0, st, p);
        tm->mark(kT_enc_compact, st);
    }
}

void launch_encode(const EncParams& p, hipStream_t st, KernelTimer* tm) {
    const int abl = p.ablate;
    if (p.channels == 3) {
        if (p.probe_xchg) launch_encode_t<3, kEncSteps, 1, 0>(p, st, tm); else launch_encode_t<3, kEncSteps, 0, 0>(p, st, tm);
        return;
    }
    if (!p.probe_xchg) { launch_encode_t<4, kEncSteps, 0, 0>(p, st, tm); return; }
    switch (abl) {   // ablation variants exist for profiling only (QOIMI_ENC_ABLATE); outputs are then invalid
        case 1: launch_encode_t<4, kEncSteps, 1, 1>(p, st, tm); break;
        case 2: launch_encode_t<4, kEncSteps, 1, 2>(p, st, tm); break;
        case 4: launch_encode_t<4, kEncSteps, 1, 4>(p, st, tm); break;
        case 7: launch_encode_t<4, kEncSteps, 1, 7>(p, st, tm); break;
        default: launch_encode_t<4, kEncSteps, 1, 0>(p, st, tm); break;
    }
}

// returns the number of mismatching patterns of the LDS exchange-order self-test (0 = ordered)
int run_lds_order_selftest(hipStream_t st) {
    uint32_t* d = nullptr;
    if (hipMalloc((void**)&d, sizeof(uint32_t)) != hipSuccess) return -1;
    (void)hipMemsetAsync(d, 0, sizeof(uint32_t), st);
    hipLaunchKernelGGL(lds_order_selftest, dim3(512), dim3(64), 0, st, d);
    uint32_t h = 1;
    if (hipMemcpyAsync(&h, d, sizeof h, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) h = 0xFFFFFFFFu;
    (void)hipFree(d);
    return (int)h;
}

}  // namespace qoimi
