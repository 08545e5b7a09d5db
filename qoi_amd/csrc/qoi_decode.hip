// qoi_decode.hip — MI355X-native QOI decoder kernels (gfx950, wave64).
//
// Replaces the sequential loop of the reference decoder (qoi.h:488-590).  The scheme —
// P1 parse, P2 slot walk, P3 symbolic summary, P4 genuine decode + exit-state check,
// restart on a failed check — is stated in qoi_decode_core.h; this file is the GPU
// plumbing around those per-chunk primitives:
//
//   * one LANE per stream segment for the P-passes.  A lane reads its segment strictly
//     sequentially, so each lane owns a small ring in LDS ([dword][lane] layout, bank =
//     lane) that is topped up with 16-byte global loads on a wave-uniform schedule (issued
//     one period ahead, so their latency hides behind the chunk arithmetic); chunks are
//     cracked out of a 64-bit register window.  HBM sees every stream byte once per pass
//     in 16-byte pieces instead of one divergent byte load per chunk.
//   * decoded pixels are collected per lane in a 16-pixel LDS row and leave as whole
//     64-byte lines (4 x dwordx4) — not as scattered dwords.
//   * the per-image chains S1 (parse), S2 (slot) and S3 (state) are two-level: 64-segment
//     groups are summarised in parallel, one wavefront chains the group summaries of an
//     image, then the groups are swept in parallel again.
#include "qoi_dev.h"
#include "qoi_kernels.h"
#include "qoi_decode_core.h"

namespace qoimi {

constexpr uint32_t kGrp = 64;      // segments per group of the two-level chains

// locate (image, segment-in-image) of global segment q / group G: images are few thousand at most
__device__ __forceinline__ uint32_t find_image(const DecImage* __restrict__ im, uint32_t n_images, uint32_t q) {
    uint32_t lo = 0, hi = n_images;            // invariant: seg_base[lo] <= q < seg_base[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (im[mid].seg_base <= q) lo = mid; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ uint32_t find_image_by_group(const DecImage* __restrict__ im, uint32_t n_images, uint32_t G) {
    uint32_t lo = 0, hi = n_images;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (im[mid].grp_base <= G) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------
// LaneReader: sequential byte source of one lane (see file header).
// Ring of RD dwords per lane; consumption is at most 5 bytes per step (longest chunk,
// qoi.h:552-557), refill() must be called by the whole wavefront every kPeriod steps.
// ---------------------------------------------------------------------------------
struct LaneReader {
    static constexpr uint32_t RD = 16;          // ring dwords per lane
    static constexpr uint32_t kPeriod = 4;      // steps between refills: <= 20 bytes consumed, 32 fetched
    uint32_t* ring;            // &lds[0][lane]; dword k of the ring at ring[k * 64]
    const uint8_t* abase;      // 16-byte aligned start of the fetched range
    const uint8_t* aend;       // first byte that must not be read (stream + size)
    uint32_t rd, wr;           // dwords pulled from / written to the ring, counted from abase
    u64 win; uint32_t nv;      // register window: next nv (>= 5) stream bytes
    uint4 pend0, pend1; uint32_t npend;

    __device__ __forceinline__ uint4 load16(const uint8_t* p) const {
        // an aligned 16-byte granule that starts inside the stream never crosses a page
        return p < aend ? *reinterpret_cast<const uint4*>(p) : make_uint4(0u, 0u, 0u, 0u);
    }
    __device__ __forceinline__ void put4(uint32_t at, const uint4& v) {
        ring[((at + 0u) & (RD - 1u)) * 64u] = v.x; ring[((at + 1u) & (RD - 1u)) * 64u] = v.y;
        ring[((at + 2u) & (RD - 1u)) * 64u] = v.z; ring[((at + 3u) & (RD - 1u)) * 64u] = v.w;
    }
    __device__ __forceinline__ void pull() {
        const uint32_t d = ring[(rd & (RD - 1u)) * 64u];
        win |= (u64)d << (8u * nv);
        nv += 4u; ++rd;
    }
    __device__ __forceinline__ void top_up() { if (nv <= 4u) pull(); if (nv <= 4u) pull(); }

    __device__ __forceinline__ void init(uint32_t* lds_col, const uint8_t* stream, uint32_t pos0, uint32_t size) {
        ring = lds_col;
        const uint8_t* p = stream + pos0;
        abase = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)15);
        aend = stream + size;
#pragma unroll
        for (uint32_t r = 0; r < RD / 4u; ++r) put4(4u * r, load16(abase + 16u * r));
        wr = RD; npend = 0;
        const uint32_t skip = (uint32_t)(p - abase);
        rd = skip >> 2;
        const uint32_t d = ring[rd * 64u]; ++rd;
        win = (u64)(d >> (8u * (skip & 3u))); nv = 4u - (skip & 3u);
        top_up();
    }
    __device__ __forceinline__ u64 peek() const { return win; }
    __device__ __forceinline__ void advance(uint32_t n) { win >>= 8u * n; nv -= n; top_up(); }
    // wave-uniform call: land the loads issued one period ago, issue the next ones
    __device__ __forceinline__ void refill() {
        if (npend > 0u) { put4(wr, pend0); wr += 4u; }
        if (npend > 1u) { put4(wr, pend1); wr += 4u; }
        const uint32_t space = RD - (wr - rd);
        npend = min(2u, space >> 2);
        if (npend > 0u) pend0 = load16(abase + (size_t)wr * 4u);
        if (npend > 1u) pend1 = load16(abase + (size_t)wr * 4u + 16u);
    }
};

// ---------------------------------------------------------------------------------
// LaneWriter: pixel sink of one lane; pixels are gathered per 16-pixel aligned group in LDS
// ([k][lane] layout) and leave as whole 64-byte (OCH 4) / 48-byte (OCH 3) lines.
// ---------------------------------------------------------------------------------
template <int OCH>
struct LaneWriter {
    uint32_t* buf;        // &lds[0][lane]; pixel k of the group at buf[k * 64]
    uint8_t* out;
    uint32_t ppos;        // next pixel index
    uint32_t gstart;      // first pixel of the current group owned by this lane (head of a segment)

    __device__ __forceinline__ void init(uint32_t* lds_col, uint8_t* image, uint32_t px_pos) {
        buf = lds_col; out = image; ppos = px_pos; gstart = px_pos & 15u;
    }
    __device__ __forceinline__ void flush(uint32_t base, uint32_t hi) {     // pixels [base+gstart, base+hi)
        if (gstart == 0u && hi == 16u) {
            uint32_t v[16];
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = buf[k * 64];
            if (OCH == 4) {
                uint4* d = reinterpret_cast<uint4*>(out + (size_t)base * 4u);
#pragma unroll
                for (int k = 0; k < 4; ++k) d[k] = make_uint4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
            } else {
                uint4* d = reinterpret_cast<uint4*>(out + (size_t)base * 3u);  // 48*(base/16): 16-byte aligned
                uint32_t w[12];
#pragma unroll
                for (int k = 0; k < 4; ++k) {                  // 4 pixels -> 3 dwords of packed r,g,b
                    const uint32_t a = v[4 * k] & 0xFFFFFFu, b = v[4 * k + 1] & 0xFFFFFFu, c = v[4 * k + 2] & 0xFFFFFFu, e = v[4 * k + 3] & 0xFFFFFFu;
                    w[3 * k] = a | (b << 24); w[3 * k + 1] = (b >> 8) | (c << 16); w[3 * k + 2] = (c >> 16) | (e << 8);
                }
#pragma unroll
                for (int k = 0; k < 3; ++k) d[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
            }
        } else {
            for (uint32_t k = gstart; k < hi; ++k) {
                const uint32_t px = buf[k * 64u];
                if (OCH == 4) reinterpret_cast<uint32_t*>(out)[base + k] = px;
                else { uint8_t* d = out + (size_t)(base + k) * 3u; d[0] = (uint8_t)px; d[1] = (uint8_t)(px >> 8); d[2] = (uint8_t)(px >> 16); }
            }
        }
        gstart = 0u;
    }
    __device__ __forceinline__ void put(uint32_t px) {
        buf[(ppos & 15u) * 64u] = px;
        ++ppos;
        if ((ppos & 15u) == 0u) flush(ppos - 16u, 16u);
    }
    __device__ __forceinline__ void finish() {
        const uint32_t hi = ppos & 15u;
        if (hi > gstart) flush(ppos & ~15u, hi);
    }
};

// ---------------------------------------------------------------------------------
// P1: parse summaries (lane = segment)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_parse(DecParams p) {
    __shared__ uint32_t s_ring[4][LaneReader::RD * 64];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const bool have = q < p.total_segs;
    const uint32_t img = find_image(p.images, p.n_images, have ? q : 0u);
    const DecImage im = p.images[img];
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    const uint32_t base = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t end = min(base + p.seg_bytes, im.chunks_end);
    const uint8_t* stream = p.streams + im.stream_off;
    LaneReader R;
    R.init(&s_ring[wave][lane], stream, base, im.chunks_end + kTrailerBytes);
    ParseState s; parse_init(s, base);
    uint32_t m = base;
    bool active = have && m < end;
    for (uint32_t it = 0; __ballot(active); ++it) {
        if ((it & (LaneReader::kPeriod - 1u)) == 0u) R.refill();
        if (active) {
            parse_step(s, m, (uint32_t)R.peek() & 0xFFu);
            const uint32_t m2 = parse_front(s);
            R.advance(m2 - m);
            m = m2;
            active = m < end;
        }
    }
    if (have) { ParseRec r; parse_finish(s, base, p.seg_bytes, r); p.parse[q] = r; }
}

// ---------------------------------------------------------------------------------
// S1: entry phase / pixel offset of every segment.  Exact.  Two-level over 64-segment groups.
//   l1  per group: ParseRec of the group (exit phase + pixels for the 5 entry phases)
//   l2  per image: chain the group records -> entry phase / pixel offset of every group
//   l3  per group: chain inside the group
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t sel5(uint32_t ph, uint32_t a0, uint32_t a1, uint32_t a2, uint32_t a3, uint32_t a4) {
    return ph == 0u ? a0 : ph == 1u ? a1 : ph == 2u ? a2 : ph == 3u ? a3 : a4;
}

__global__ __launch_bounds__(64) void dec_chain_parse_l1(DecParams p) {
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    const uint32_t cnt = min(kGrp, im.nseg - j0);
    ParseRec r; r.exit_phase = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) r.pixels[k] = 0;
    if (lane < cnt) r = p.parse[im.seg_base + j0 + lane];
    // lanes 0..4 walk the group for entry phase = lane; records are broadcast from their lanes
    uint32_t ph = min(lane, 4u);
    u64 sum = 0;
    for (uint32_t l = 0; l < cnt; ++l) {
        const uint32_t ex = read_lane_dyn(r.exit_phase, l);
        const uint32_t a0 = read_lane_dyn(r.pixels[0], l), a1 = read_lane_dyn(r.pixels[1], l), a2 = read_lane_dyn(r.pixels[2], l),
                       a3 = read_lane_dyn(r.pixels[3], l), a4 = read_lane_dyn(r.pixels[4], l);
        sum += sel5(ph, a0, a1, a2, a3, a4);
        ph = (ex >> (3u * ph)) & 7u;
    }
    const uint32_t capped = sum > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)sum;   // > any legal pixel count anyway
    // gather the five lanes' results into one record
    ParseRec g;
    g.exit_phase = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) { g.exit_phase |= read_lane(ph, k) << (3 * k); g.pixels[k] = read_lane(capped, k); }
    if (lane == 0) p.grp_parse[G] = g;
}

__global__ __launch_bounds__(64) void dec_chain_parse_l2(DecParams p) {
    const uint32_t img = blockIdx.x, lane = lane_id();
    const DecImage im = p.images[img];
    uint32_t phase = 0; u64 off = 0;
    for (uint32_t g0 = 0; g0 < im.ngrp; g0 += 64u) {
        ParseRec r; r.exit_phase = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) r.pixels[k] = 0;
        if (g0 + lane < im.ngrp) r = p.grp_parse[im.grp_base + g0 + lane];
        uint32_t my_phase = 0, my_off = 0;
        const uint32_t cnt = min(64u, im.ngrp - g0);
        for (uint32_t l = 0; l < cnt; ++l) {
            if (lane == l) { my_phase = phase; my_off = (uint32_t)min(off, (u64)im.npx); }
            const uint32_t add = read_lane_dyn(sel5(phase, r.pixels[0], r.pixels[1], r.pixels[2], r.pixels[3], r.pixels[4]), l);
            const uint32_t ex = read_lane_dyn(r.exit_phase, l);
            off = min(off + add, (u64)im.npx);
            phase = (ex >> (3u * phase)) & 7u;
        }
        if (g0 + lane < im.ngrp) { p.grp_phase[im.grp_base + g0 + lane] = (uint8_t)my_phase; p.grp_off[im.grp_base + g0 + lane] = my_off; }
    }
    if (lane == 0) {
        p.images[img].total_px = (uint32_t)off;
        p.images[img].n_active = 0;              // raised by l3
        p.images[img].start_seg = 0;
        p.first_bad[img] = 0xFFFFFFFFu;
    }
}

__global__ __launch_bounds__(64) void dec_chain_parse_l3(DecParams p) {
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    const uint32_t cnt = min(kGrp, im.nseg - j0);
    ParseRec r; r.exit_phase = 0;
#pragma unroll
    for (int k = 0; k < 5; ++k) r.pixels[k] = 0;
    if (lane < cnt) r = p.parse[im.seg_base + j0 + lane];
    uint32_t phase = p.grp_phase[G]; u64 off = p.grp_off[G];
    uint32_t my_phase = 0, my_off = 0, n_active = 0;     // n_active: segments that start before the pixel limit
    for (uint32_t l = 0; l < cnt; ++l) {
        if (lane == l) { my_phase = phase; my_off = (uint32_t)off; }
        if (off < im.npx) n_active = j0 + l + 1u;
        const uint32_t add = read_lane_dyn(sel5(phase, r.pixels[0], r.pixels[1], r.pixels[2], r.pixels[3], r.pixels[4]), l);
        const uint32_t ex = read_lane_dyn(r.exit_phase, l);
        off = min(off + add, (u64)im.npx);
        phase = (ex >> (3u * phase)) & 7u;
    }
    if (lane < cnt) { p.entry_phase[im.seg_base + j0 + lane] = (uint8_t)my_phase; p.px_off[im.seg_base + j0 + lane] = my_off; }
    if (lane == 0 && n_active) atomicMax(&p.images[img].n_active, n_active);
}

// ---------------------------------------------------------------------------------
// P2: speculative slot/alpha transfer (lane = segment)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_slot_walk(DecParams p) {
    __shared__ uint32_t s_ring[4][LaneReader::RD * 64];
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    bool have = q < p.total_segs;
    const uint32_t img = find_image(p.images, p.n_images, have ? q : 0u);
    const DecImage im = p.images[img];
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    have = have && j >= im.start_seg && j < im.n_active;
    if (!__ballot(have)) return;
    const uint32_t base = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t end = min(base + p.seg_bytes, im.chunks_end);
    uint32_t pos = base + (have ? p.entry_phase[q] : 0u);
    LaneReader R;
    R.init(&s_ring[wave][lane], p.streams + im.stream_off, pos, im.chunks_end + kTrailerBytes);
    SlotState s; slot_init(s);
    bool active = have && pos < end;
    for (uint32_t it = 0; __ballot(active); ++it) {
        if ((it & (LaneReader::kPeriod - 1u)) == 0u) R.refill();
        if (active) {
            const Chunk c = crack(R.peek());
            slot_step(s, c);
            R.advance(c.len);
            pos += c.len;
            active = pos < end;
        }
    }
    if (have) { SlotRec r; slot_finish(s, r); p.slot_rec[q] = r; }
}

__device__ __forceinline__ uint32_t slot_pack(const SlotRec& r) {
    return r.hc | ((uint32_t)r.h_rel << 8) | ((uint32_t)r.h_alpha << 9) | ((uint32_t)r.a_abs << 10) | ((uint32_t)r.ac << 16);
}
__device__ __forceinline__ SlotRec slot_unpack(uint32_t w) {
    SlotRec t; t.hc = w & 63u; t.h_rel = (w >> 8) & 1u; t.h_alpha = (w >> 9) & 1u; t.a_abs = (w >> 10) & 1u; t.ac = (w >> 16) & 0xFFu;
    return t;
}

// S2 l1: compose the transfers of the group's segments that are still to be decoded
__global__ __launch_bounds__(64) void dec_chain_slots_l1(DecParams p) {
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 + kGrp <= im.start_seg || j0 >= im.n_active) return;
    const uint32_t lo = max(j0, im.start_seg), hi = min(j0 + kGrp, im.n_active);
    SlotRec mine = {0, 1, 0, 0, 0};
    if (j0 + lane >= lo && j0 + lane < hi) mine = p.slot_rec[im.seg_base + j0 + lane];
    const uint32_t packed = slot_pack(mine);
    SlotRec acc = {0, 1, 0, 0, 0};
    for (uint32_t l = lo - j0; l < hi - j0; ++l) acc = slot_compose(acc, slot_unpack(read_lane_dyn(packed, l)));
    if (lane == 0) p.grp_slot[G] = acc;
}

// S2 l2: chain the groups of one image from the group holding start_seg; the start value is the
// hash/alpha of start_seg's concrete entry pixel
__global__ __launch_bounds__(64) void dec_chain_slots_l2(DecParams p) {
    const uint32_t img = blockIdx.x, lane = lane_id();
    const DecImage im = p.images[img];
    if (im.start_seg >= im.n_active) return;
    const uint32_t px0 = p.entry[(size_t)(im.seg_base + im.start_seg) * 65u + 64u];
    uint32_t slot = hash_px(px0), alpha = px0 >> 24;
    const uint32_t gfirst = im.start_seg / kGrp, gend = (im.n_active + kGrp - 1u) / kGrp;
    for (uint32_t g0 = gfirst; g0 < gend; g0 += 64u) {
        SlotRec mine = {0, 1, 0, 0, 0};
        if (g0 + lane < gend) mine = p.grp_slot[im.grp_base + g0 + lane];
        const uint32_t packed = slot_pack(mine);
        uint32_t my_slot = 0, my_alpha = 0;
        const uint32_t cnt = min(64u, gend - g0);
        for (uint32_t l = 0; l < cnt; ++l) {
            if (lane == l) { my_slot = slot; my_alpha = alpha; }
            slot_apply(slot_unpack(read_lane_dyn(packed, l)), slot, alpha);
        }
        if (g0 + lane < gend) { p.grp_slot_in[im.grp_base + g0 + lane] = (uint8_t)my_slot; p.grp_alpha_in[im.grp_base + g0 + lane] = (uint8_t)my_alpha; }
    }
}

// S2 l3: sweep inside every group
__global__ __launch_bounds__(64) void dec_chain_slots_l3(DecParams p) {
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 + kGrp <= im.start_seg || j0 >= im.n_active) return;
    const uint32_t lo = max(j0, im.start_seg), hi = min(j0 + kGrp, im.n_active);
    SlotRec mine = {0, 1, 0, 0, 0};
    if (j0 + lane >= lo && j0 + lane < hi) mine = p.slot_rec[im.seg_base + j0 + lane];
    const uint32_t packed = slot_pack(mine);
    uint32_t slot = p.grp_slot_in[G], alpha = p.grp_alpha_in[G];
    uint32_t my_slot = 0, my_alpha = 0;
    for (uint32_t l = lo - j0; l < hi - j0; ++l) {
        if (lane == l) { my_slot = slot; my_alpha = alpha; }
        slot_apply(slot_unpack(read_lane_dyn(packed, l)), slot, alpha);
    }
    if (j0 + lane >= lo && j0 + lane < hi) { p.slot_in[im.seg_base + j0 + lane] = (uint8_t)my_slot; p.alpha_in[im.seg_base + j0 + lane] = (uint8_t)my_alpha; }
}

// ---------------------------------------------------------------------------------
// P3: symbolic summaries (lane = segment; private 64-entry symbolic table in LDS,
// laid out [slot][lane] so a lane always hits its own bank pair)
// ---------------------------------------------------------------------------------
struct LdsSymTab {
    sym_t* col;   // &lds[0][lane]
    __device__ __forceinline__ sym_t get(uint32_t k) const { return col[k * 64u]; }
    __device__ __forceinline__ void set(uint32_t k, sym_t v) { col[k * 64u] = v; }
};

__global__ __launch_bounds__(64) void dec_summarize(DecParams p) {
    __shared__ sym_t s_tab[64 * 64];
    __shared__ uint32_t s_ring[LaneReader::RD * 64];
    const uint32_t lane = lane_id();
    const uint32_t q = blockIdx.x * 64u + lane;
    bool have = q < p.total_segs;
    const uint32_t img = find_image(p.images, p.n_images, have ? q : 0u);
    const DecImage im = p.images[img];
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    have = have && j >= im.start_seg && j < im.n_active;
    if (!__ballot(have)) return;
    const uint32_t base = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t end = min(base + p.seg_bytes, im.chunks_end);
    uint32_t pos = base + (have ? p.entry_phase[q] : 0u);
    LaneReader R;
    R.init(&s_ring[lane], p.streams + im.stream_off, pos, im.chunks_end + kTrailerBytes);
    LdsSymTab tab{&s_tab[lane]};
    SymState s; sym_init(s, have ? p.slot_in[q] : 0u, have ? p.alpha_in[q] : 0u, tab);
    bool active = have && pos < end;
    for (uint32_t it = 0; __ballot(active); ++it) {
        if ((it & (LaneReader::kPeriod - 1u)) == 0u) R.refill();
        if (active) {
            const Chunk c = crack(R.peek());
            sym_step(s, c, tab);
            R.advance(c.len);
            pos += c.len;
            active = pos < end;
        }
    }
    if (have) {
        sym_t* dst = p.summary + (size_t)q * 65u;
        for (uint32_t k = 0; k < 64u; ++k) dst[k] = tab.get(k);
        dst[64] = sym_pixel(s);
    }
}

// ---------------------------------------------------------------------------------
// S3: concrete (px, index[64]) at every segment entry.  lane = table slot; the pixel word is
// kept redundantly by every lane.
//   l1  per group: compose the segments' symbolic summaries into the group's summary
//   l2  per image: apply the group summaries in sequence to the concrete state
//   l3  per group: apply the segment summaries in sequence from the group's entry state
// ---------------------------------------------------------------------------------
__device__ __forceinline__ sym_t gather_sym(sym_t tabv, sym_t pxv, uint32_t src) {
    const uint32_t lo = gather_lane((uint32_t)tabv, src & 63u), hi = gather_lane((uint32_t)(tabv >> 32), src & 63u);
    return src == 64u ? pxv : ((sym_t)lo | ((sym_t)hi << 32));
}

__global__ __launch_bounds__(64) void dec_chain_state_l1(DecParams p) {
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 + kGrp <= im.start_seg || j0 >= im.n_active) return;
    const uint32_t lo = max(j0, im.start_seg), hi = min(j0 + kGrp, im.n_active);
    sym_t P_tab = sym_make(0u, lane, 0u), P_px = sym_make(0u, 64u, 0u);          // identity
    const sym_t* __restrict__ sum = p.summary + (size_t)(im.seg_base + lo) * 65u;
    sym_t s_tab = sum[lane], s_px = sum[64];
    for (uint32_t j = lo; j < hi; ++j) {
        const sym_t c_tab = s_tab, c_px = s_px;
        if (j + 1u < hi) { s_tab = sum[(size_t)(j + 1u - lo) * 65u + lane]; s_px = sum[(size_t)(j + 1u - lo) * 65u + 64u]; }   // prefetch
        const sym_t n_tab = sym_compose(c_tab, gather_sym(P_tab, P_px, sym_src(c_tab)));
        const sym_t n_px = sym_compose(c_px, gather_sym(P_tab, P_px, sym_src(c_px)));
        P_tab = n_tab; P_px = n_px;
    }
    p.grp_summary[(size_t)G * 65u + lane] = P_tab;
    if (lane == 0) p.grp_summary[(size_t)G * 65u + 64u] = P_px;
}

__global__ __launch_bounds__(64) void dec_chain_state_l2(DecParams p) {
    const uint32_t img = blockIdx.x, lane = lane_id();
    const DecImage im = p.images[img];
    if (im.start_seg >= im.n_active) return;
    const size_t q0 = (size_t)im.seg_base + im.start_seg;
    uint32_t tabv = p.entry[q0 * 65u + lane];        // concrete entry state of start_seg is given
    uint32_t pxv = p.entry[q0 * 65u + 64u];
    const uint32_t gfirst = im.start_seg / kGrp, gend = (im.n_active + kGrp - 1u) / kGrp;
    const sym_t* __restrict__ gs = p.grp_summary + (size_t)(im.grp_base + gfirst) * 65u;
    sym_t s_tab = gs[lane], s_px = gs[64];
    for (uint32_t g = gfirst; g < gend; ++g) {
        const size_t G = (size_t)im.grp_base + g;
        p.grp_entry[G * 65u + lane] = tabv;
        if (lane == 0) p.grp_entry[G * 65u + 64u] = pxv;
        const sym_t c_tab = s_tab, c_px = s_px;
        if (g + 1u < gend) { s_tab = gs[(size_t)(g + 1u - gfirst) * 65u + lane]; s_px = gs[(size_t)(g + 1u - gfirst) * 65u + 64u]; }
        const uint32_t src_t = sym_src(c_tab), src_p = sym_src(c_px);
        const uint32_t g_t = gather_lane(tabv, src_t & 63u), g_p = gather_lane(tabv, src_p & 63u);
        const uint32_t ntab = sym_eval(c_tab, src_t == 64u ? pxv : g_t);
        const uint32_t npx = sym_eval(c_px, src_p == 64u ? pxv : g_p);
        tabv = ntab; pxv = npx;
    }
}

__global__ __launch_bounds__(64) void dec_chain_state_l3(DecParams p) {
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 + kGrp <= im.start_seg || j0 >= im.n_active) return;
    const uint32_t lo = max(j0, im.start_seg), hi = min(j0 + kGrp, im.n_active);
    uint32_t tabv = p.grp_entry[(size_t)G * 65u + lane], pxv = p.grp_entry[(size_t)G * 65u + 64u];
    const sym_t* __restrict__ sum = p.summary + (size_t)(im.seg_base + lo) * 65u;
    uint32_t* __restrict__ ent = p.entry + (size_t)(im.seg_base + lo) * 65u;
    sym_t s_tab = sum[lane], s_px = sum[64];
    for (uint32_t j = lo; j < hi; ++j) {
        if (j > im.start_seg) {                       // start_seg's entry state is given, never rewritten
            ent[(size_t)(j - lo) * 65u + lane] = tabv;
            if (lane == 0) ent[(size_t)(j - lo) * 65u + 64u] = pxv;
        }
        const sym_t c_tab = s_tab, c_px = s_px;
        if (j + 1u < hi) { s_tab = sum[(size_t)(j + 1u - lo) * 65u + lane]; s_px = sum[(size_t)(j + 1u - lo) * 65u + 64u]; }
        const uint32_t src_t = sym_src(c_tab), src_p = sym_src(c_px);
        const uint32_t g_t = gather_lane(tabv, src_t & 63u), g_p = gather_lane(tabv, src_p & 63u);
        const uint32_t ntab = sym_eval(c_tab, src_t == 64u ? pxv : g_t);
        const uint32_t npx = sym_eval(c_px, src_p == 64u ? pxv : g_p);
        tabv = ntab; pxv = npx;
    }
}

// ---------------------------------------------------------------------------------
// P4: genuine decode of every active segment + exit-state check (lane = segment)
// ---------------------------------------------------------------------------------
struct LdsTab32 {
    uint32_t* col;
    __device__ __forceinline__ uint32_t get(uint32_t k) const { return col[k * 64u]; }
    __device__ __forceinline__ void set(uint32_t k, uint32_t v) { col[k * 64u] = v; }
};

template <int OCH>
__global__ __launch_bounds__(64) void dec_segments(DecParams p) {
    __shared__ uint32_t s_tab[64 * 64];
    __shared__ uint32_t s_ring[LaneReader::RD * 64];
    __shared__ uint32_t s_out[16 * 64];
    const uint32_t lane = lane_id();
    const uint32_t q = blockIdx.x * 64u + lane;
    bool have = q < p.total_segs;
    const uint32_t img = find_image(p.images, p.n_images, have ? q : 0u);
    const DecImage im = p.images[img];
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    have = have && j >= im.start_seg && j < im.n_active;
    if (!__ballot(have)) return;
    const uint32_t base = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t end = min(base + p.seg_bytes, im.chunks_end);
    uint32_t pos = base + (have ? p.entry_phase[q] : 0u);
    LaneReader R;
    R.init(&s_ring[lane], p.streams + im.stream_off, pos, im.chunks_end + kTrailerBytes);
    LaneWriter<OCH> W;
    W.init(&s_out[lane], p.pixels + (size_t)img * p.pixel_stride, have ? p.px_off[q] : 0u);
    LdsTab32 tab{&s_tab[lane]};
    const uint32_t* __restrict__ ent = p.entry + (size_t)(have ? q : 0u) * 65u;
    uint32_t px = 0;
    if (have) {
        for (uint32_t k = 0; k < 64u; ++k) tab.set(k, ent[k]);
        px = ent[64];
    }
    const uint32_t limit = im.npx;
    bool active = have && pos < end && W.ppos < limit;
    for (uint32_t it = 0; __ballot(active); ++it) {
        if ((it & (LaneReader::kPeriod - 1u)) == 0u) R.refill();
        if (active) {
            const Chunk c = crack(R.peek());
            px = pixel_step(px, c, tab);
            const uint32_t n = min(chunk_run(c), limit - W.ppos);      // over-long run clipped (Appendix B item 8)
            for (uint32_t k = 0; k < n; ++k) W.put(px);
            R.advance(c.len);
            pos += c.len;
            active = pos < end && W.ppos < limit;
        }
    }
    if (have) {
        W.finish();
        if (j + 1u < im.n_active) {
            // exit state must equal what the next segment was started from
            const uint32_t* __restrict__ nxt = ent + 65u;
            bool same = nxt[64] == px;
            for (uint32_t k = 0; k < 64u; ++k) same = same && (nxt[k] == tab.get(k));
            if (!same) {
                uint32_t* fx = p.fix + (size_t)(q + 1u) * 65u;
                for (uint32_t k = 0; k < 64u; ++k) fx[k] = tab.get(k);
                fx[64] = px;
                atomicMin(&p.first_bad[img], j + 1u);
            }
        } else {
            p.images[img].final_px = px;     // pixel repeated when the stream ends early (qoi.h:544)
        }
    }
}

// Pixels the chunks never reach repeat the last pixel (truncated streams, size==22).
template <int OCH>
__global__ __launch_bounds__(256) void dec_fill(DecParams p) {
    const uint32_t img = blockIdx.y;
    const DecImage im = p.images[img];
    const uint32_t px = im.n_active ? im.final_px : kInitPx;
    uint8_t* out = p.pixels + (size_t)img * p.pixel_stride;
    for (uint32_t i = im.total_px + blockIdx.x * 256u + threadIdx.x; i < im.npx; i += gridDim.x * 256u) {
        if (OCH == 4) reinterpret_cast<uint32_t*>(out)[i] = px;
        else { uint8_t* d = out + (size_t)i * 3u; d[0] = (uint8_t)px; d[1] = (uint8_t)(px >> 8); d[2] = (uint8_t)(px >> 16); }
    }
}

// Concrete start state of every image: {0,0,0,255} and a zeroed table (qoi.h:533-537).
__global__ __launch_bounds__(64) void dec_init_state(DecParams p) {
    const uint32_t img = blockIdx.x, lane = lane_id();
    const DecImage im = p.images[img];
    if (im.nseg == 0) return;
    p.entry[(size_t)im.seg_base * 65u + lane] = 0u;
    if (lane == 0) p.entry[(size_t)im.seg_base * 65u + 64u] = kInitPx;
}

// After a round: images whose check failed restart at the first bad segment from the
// TRUE exit state of its predecessor; the others are finished.  Counts pending images.
__global__ __launch_bounds__(64) void dec_prepare_restart(DecParams p) {
    const uint32_t img = blockIdx.x, lane = lane_id();
    const DecImage im = p.images[img];
    const uint32_t fb = p.first_bad[img];
    if (fb == 0xFFFFFFFFu) {
        if (lane == 0) p.images[img].start_seg = im.n_active;    // done
        return;
    }
    const size_t q = (size_t)im.seg_base + fb;
    p.entry[q * 65u + lane] = p.fix[q * 65u + lane];
    if (lane == 0) {
        p.entry[q * 65u + 64u] = p.fix[q * 65u + 64u];
        p.images[img].start_seg = fb;
        p.first_bad[img] = 0xFFFFFFFFu;
        atomicAdd(p.pending, 1u);
        atomicAdd(p.redo_segs, im.n_active - fb);
    }
}

// ---------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------
void launch_decode_parse(const DecParams& p, hipStream_t st, KernelTimer* tm) {
    tm->mark(kT_begin, st);
    if (p.total_segs) {
        hipLaunchKernelGGL(dec_parse, dim3((p.total_segs + 255u) / 256u), dim3(256), 0, st, p);
        tm->mark(kT_dec_parse, st);
        hipLaunchKernelGGL(dec_chain_parse_l1, dim3(p.total_grps), dim3(64), 0, st, p);
    }
    hipLaunchKernelGGL(dec_chain_parse_l2, dim3(p.n_images), dim3(64), 0, st, p);
    if (p.total_segs) hipLaunchKernelGGL(dec_chain_parse_l3, dim3(p.total_grps), dim3(64), 0, st, p);
    hipLaunchKernelGGL(dec_init_state, dim3(p.n_images), dim3(64), 0, st, p);
    tm->mark(kT_dec_chain_parse, st);
}

void launch_decode_round(const DecParams& p, int out_channels, hipStream_t st, KernelTimer* tm) {
    if (!p.total_segs) return;
    const uint32_t b256 = (p.total_segs + 255u) / 256u, b64 = (p.total_segs + 63u) / 64u;
    tm->mark(kT_begin, st);
    hipLaunchKernelGGL(dec_slot_walk, dim3(b256), dim3(256), 0, st, p);
    tm->mark(kT_dec_slot_walk, st);
    hipLaunchKernelGGL(dec_chain_slots_l1, dim3(p.total_grps), dim3(64), 0, st, p);
    hipLaunchKernelGGL(dec_chain_slots_l2, dim3(p.n_images), dim3(64), 0, st, p);
    hipLaunchKernelGGL(dec_chain_slots_l3, dim3(p.total_grps), dim3(64), 0, st, p);
    tm->mark(kT_dec_chain_slots, st);
    hipLaunchKernelGGL(dec_summarize, dim3(b64), dim3(64), 0, st, p);
    tm->mark(kT_dec_summarize, st);
    hipLaunchKernelGGL(dec_chain_state_l1, dim3(p.total_grps), dim3(64), 0, st, p);
    hipLaunchKernelGGL(dec_chain_state_l2, dim3(p.n_images), dim3(64), 0, st, p);
    hipLaunchKernelGGL(dec_chain_state_l3, dim3(p.total_grps), dim3(64), 0, st, p);
    tm->mark(kT_dec_chain_state, st);
    if (out_channels == 4) hipLaunchKernelGGL(dec_segments<4>, dim3(b64), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(dec_segments<3>, dim3(b64), dim3(64), 0, st, p);
    tm->mark(kT_dec_segments, st);
    hipLaunchKernelGGL(dec_prepare_restart, dim3(p.n_images), dim3(64), 0, st, p);
    tm->mark(kT_dec_restart, st);
}

void launch_decode_fill(const DecParams& p, int out_channels, hipStream_t st, KernelTimer* tm) {
    const dim3 grid(256, p.n_images);
    tm->mark(kT_begin, st);
    if (out_channels == 4) hipLaunchKernelGGL(dec_fill<4>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(dec_fill<3>, grid, dim3(256), 0, st, p);
    tm->mark(kT_dec_fill, st);
}

}  // namespace qoimi
