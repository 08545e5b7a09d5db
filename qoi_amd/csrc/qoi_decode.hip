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
#include <type_traits>
#include "qoi_kernels.h"
#include "qoi_decode_core.h"

namespace qoimi {

constexpr uint32_t kGrp = 64;      // segments per group of the two-level chains
constexpr uint32_t kSummaryDescs = 32;   // run descriptors (16 bytes) that fit a segment's slot of symbolic summaries (65 x 8 bytes)

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
// LaneReader: byte source of one lane (see file header).  Ring of RD dwords per lane in LDS
// ([dword][lane] layout, bank = lane) plus one mirror dword (slot RD repeats slot 0), so the two
// dwords that hold a chunk's first bytes are always at ring[k] and ring[k + 1]: a chunk is read with
// ONE ds_read2_b32 at its byte position and one v_alignbit - no shifting window, no per-chunk
// top-up.  A step consumes at most 5 bytes (longest chunk, qoi.h:552-557); refill() must be
// called by the whole wavefront every kPeriod steps: it lands the 16-byte loads issued one
// period earlier (their latency hides behind the chunk arithmetic) and issues the next ones.
// ---------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) uint32_t lds_u32;
// Stream bytes are read through GLOBAL-address-space pointers.  Left generic, hipcc emits flat_load here (the
// pointer travels through struct members and loop phis); a FLAT load in flight counts on both vmcnt and lgkmcnt
// and may return out of order with LDS data, so every wait for an LDS read would also drain the prefetch loads.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const u32x4 gconst_u32x4;
__device__ __forceinline__ uint4 load_global16(const uint8_t* p) {
    const u32x4 v = *(gconst_u32x4*)(uintptr_t)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ uint32_t lds_addr_of(const void* p) { return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) char*)p; }

// RD_: ring dwords per lane (power of two).  NP_: 16-byte loads a lane keeps in flight.  PERIOD_: steps
// between refills.  The passes are bound by the latency of these per-lane streams (a lane consumes ~1.2
// bytes per step; with 32 bytes in flight it waits for memory every few steps), so bytes in flight per
// lane is the number that matters: ring 128 B + 64 B in flight here.  Invariant (dwords, o = landed and
// unread): a period consumes <= 5*PERIOD_ bytes (+5 for the one-chunk look-ahead of the callers), a refill
// tops the ring up in 16-byte pieces; with RD_=32, NP_=4, PERIOD_=8: o >= 17 after every refill, so the 8 bytes
// a peek reads are always there.
// LINE_: the ring is topped up only by whole, 64-byte aligned lines (NP_ x 16 bytes issued together, or nothing).  With many
// long-lived lanes per CU (dec_transcode: 512) a lane's 16-byte loads come too far apart for its cache line to still be in the
// L2 the next time (32 K lane streams per XCD on 32 K L2 lines): every line was fetched from HBM several times.
// DELAY_ (with LINE_): the loads land DELAY_ periods after the period that follows their issue - at most one line is in
// flight, and it gets 2 * PERIOD_ steps (about a microsecond) to arrive before a wavefront waits for it.
template <int RD_, int NP_, int PERIOD_, int LOOK_ = 8, bool LINE_ = false, int DELAY_ = 0>
struct LaneReaderT {
    static constexpr uint32_t RD = RD_;
    static constexpr uint32_t kSlots = RD + 1;
    static constexpr uint32_t kPeriod = PERIOD_;
    static_assert((RD_ & (RD_ - 1)) == 0 && RD_ >= 16, "ring size");
    // LOOK_: bytes a caller reads beyond its position (8: one peek; 16: the four-dword window of dec_segments)
    static_assert(4 * RD_ - 16 - 12 - (5 * PERIOD_ + 5) >= LOOK_ + 3, "ring too small for the period");
    static_assert(16 * NP_ >= 5 * PERIOD_, "refill rate below the worst-case consumption");
    uint32_t ring;             // LDS byte address of ring[0][lane]; dword k of the ring at ring + k*256
    const uint8_t* abase;      // 16-byte aligned start of the fetched range
    const uint8_t* alast;      // last 16-byte granule that starts before stream + size
    uint32_t aoff;             // abase - stream: stream position p sits at ring byte (p - aoff)
    uint32_t wr;               // dwords landed in the ring, counted from abase
    uint4 pend[NP_]; uint32_t npend;
    uint32_t age;              // DELAY_: periods the pending line has been in flight

    __device__ __forceinline__ uint4 load16(const uint8_t* p) const {
        // An aligned 16-byte granule that starts inside the stream never crosses a page.  A granule past the end is
        // replaced by the last one inside (no branch, no zero fill): those ring bytes stand for stream positions that
        // no chunk reads.
        return load_global16(p < alast ? p : alast);
    }
    __device__ __forceinline__ void put4(uint32_t at, const uint4& v) {       // at: multiple of 4
        const uint32_t a = ring + (at & (RD - 1u)) * 256u;
        lds_u32* q = (lds_u32*)a;
        q[0] = v.x; q[64] = v.y; q[128] = v.z; q[192] = v.w;
        if ((at & (RD - 1u)) == 0u) ((lds_u32*)ring)[RD * 64u] = v.x;         // mirror of slot 0
    }
    __device__ __forceinline__ void init(uint32_t ring_addr, const uint8_t* stream, uint32_t pos0, uint32_t size) {
        ring = ring_addr;
        const uint8_t* p = stream + pos0;
        abase = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)(LINE_ ? 16 * NP_ - 1 : 15));
        alast = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(stream + size - 1u) & ~(uintptr_t)15);
        aoff = pos0 - (uint32_t)(p - abase);
        if (LINE_) {                                                  // loads first, LDS writes after: RD / 4 loads in flight
            uint4 v[RD / 4u];
#pragma unroll
            for (uint32_t r = 0; r < RD / 4u; ++r) v[r] = load16(abase + 16u * r);
#pragma unroll
            for (uint32_t r = 0; r < RD / 4u; ++r) put4(4u * r, v[r]);
        } else {
#pragma unroll
        for (uint32_t r = 0; r < RD / 4u; ++r) put4(4u * r, load16(abase + 16u * r));
        }
        wr = RD; npend = 0; age = 0;
    }
    // chunk bytes 0..3 (w32) and byte 4 (b5) of the chunk at stream position pos
    __device__ __forceinline__ void peek(uint32_t pos, uint32_t& w32, uint32_t& b5) const {
        const uint32_t rp = pos - aoff;
        const lds_u32* q = (const lds_u32*)(ring + ((rp >> 2) & (RD - 1u)) * 256u);
        const uint32_t d0 = q[0], d1 = q[64];
        const uint32_t sh = (rp & 3u) * 8u;
        w32 = __builtin_amdgcn_alignbit(d1, d0, sh);
        b5 = (d1 >> sh) & 0xFFu;
    }
    // wave-uniform calls; pos = the lane's current stream position.  land(): the loads issued one period ago
    // go into the ring (the only place that waits for memory).  issue(): next loads.  Callers that STORE to
    // global memory do so right AFTER issue(): the vector-memory counter retires in order, so a store issued
    // shortly before land() would make that wait a wait for the store's acknowledgement; issued a whole
    // period earlier it has long completed.
    __device__ __forceinline__ void land() {
        if (DELAY_ > 0) {
            // wave-uniform: the oldest pending line of any lane decides (lanes issue when THEY have room; a lane whose line is
            // younger lands it a little early - it was asked for at least one period ago, see issue())
            const bool due_now = lanes_where(npend != 0u && age >= (uint32_t)DELAY_) != 0;
            if (due_now) {
#pragma unroll
                for (int i = 0; i < NP_; ++i) if ((uint32_t)i < npend) put4(wr + 4u * i, pend[i]);
                wr += 4u * npend; npend = 0u;
            }
            age += npend ? 1u : 0u;
            return;
        }
#pragma unroll
        for (int i = 0; i < NP_; ++i) if ((uint32_t)i < npend) put4(wr + 4u * i, pend[i]);
        wr += 4u * npend;
    }
    __device__ __forceinline__ void issue(uint32_t pos) {
        const uint32_t space = RD - (wr - ((pos - aoff) >> 2));
        if (DELAY_ > 0) {
            const bool go = npend == 0u && (space >> 2) >= (uint32_t)NP_;
            if (lanes_where(go)) {
#pragma unroll
                for (int i = 0; i < NP_; ++i) if (go) pend[i] = load16(abase + (size_t)wr * 4u + 16u * i);
                if (go) { npend = (uint32_t)NP_; age = 0u; }
            }
            return;
        }
        npend = LINE_ ? ((space >> 2) >= (uint32_t)NP_ ? (uint32_t)NP_ : 0u) : min((uint32_t)NP_, space >> 2);
#pragma unroll
        for (int i = 0; i < NP_; ++i) if ((uint32_t)i < npend) pend[i] = load16(abase + (size_t)wr * 4u + 16u * i);
    }
    __device__ __forceinline__ void refill(uint32_t pos) { land(); issue(pos); }
    __device__ __forceinline__ bool due(uint32_t it) const { return (it % kPeriod) == 0u; }
};
typedef LaneReaderT<16, 2, 4> LaneReader;

// chunk length in P3/P4: from the chunk-table word that was fetched a step ahead (two instructions) or by
// arithmetic on the tag byte (seven, but off the LDS round trip)
#ifdef QOIMI_LEN_ARITH
#define QOIMI_STEP_LEN(b1, info) len_of(b1)
#else
#define QOIMI_STEP_LEN(b1, info) lut_len(info)
#endif

// 256-entry chunk table (qoi_decode_core.h: lut_entry) in LDS; built by the first 256 threads / by 64 lanes x 4
struct LdsLut {
    uint32_t delta[256], info[256];
};
__device__ __forceinline__ void build_lut(LdsLut& L, uint32_t tid, uint32_t nthreads) {
    for (uint32_t b = tid; b < 256u; b += nthreads) lut_entry(b, L.delta[b], L.info[b]);
    __syncthreads();
}

// Device forms of the byte-wise pixel arithmetic: one SDWA add per channel (the result's low byte lands in
// the selected byte of acc, the other bytes are preserved) instead of a 5-op SWAR add.
__device__ __forceinline__ void add_byte0(uint32_t& acc, uint32_t b) {
    asm("v_add_u32_sdwa %0, %0, %1 dst_sel:BYTE_0 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_0 src1_sel:BYTE_0" : "+v"(acc) : "v"(b));
}
__device__ __forceinline__ void add_byte1(uint32_t& acc, uint32_t b) {
    asm("v_add_u32_sdwa %0, %0, %1 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_1 src1_sel:BYTE_1" : "+v"(acc) : "v"(b));
}
__device__ __forceinline__ void add_byte2(uint32_t& acc, uint32_t b) {
    asm("v_add_u32_sdwa %0, %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_2" : "+v"(acc) : "v"(b));
}
__device__ __forceinline__ void add_dword_byte2(uint32_t& acc, uint32_t b) {     // acc += b.byte2
    asm("v_add_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "+v"(acc) : "v"(b));
}
__device__ __forceinline__ void add_byte2_from0(uint32_t& acc, uint32_t b) {     // acc.byte2 += b.byte0
    asm("v_add_u32_sdwa %0, %0, %1 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:BYTE_2 src1_sel:BYTE_0" : "+v"(acc) : "v"(b));
}
// px + (dr,dg,db) of a relative chunk (qoi.h:561-572); delta0/info from the chunk table, w32 = chunk bytes 0..3
__device__ __forceinline__ uint32_t apply_relative(uint32_t px, uint32_t w32, uint32_t delta0, uint32_t info) {
    const uint32_t m = (uint32_t)__builtin_amdgcn_sbfe((int)info, 28, 1);        // all ones for LUMA
    const uint32_t er = __builtin_amdgcn_ubfe(w32, 12, 4) & m;                   // b2 >> 4
    const uint32_t eb = __builtin_amdgcn_ubfe(w32, 8, 4) & m;                    // b2 & 15
    uint32_t r = px;
    add_byte0(r, delta0); add_byte1(r, delta0); add_byte2(r, delta0);
    add_byte0(r, er); add_byte2_from0(r, eb);
    return r;
}

// ---------------------------------------------------------------------------------
// LaneWriter: pixel sink of one lane.  A lane produces its segment's pixels in order into a 16-pixel
// ring in LDS ([k][lane] layout, 4 KiB per wavefront); drain() writes every complete, 4-pixel aligned
// group as one 16-byte (OCH 4) / 12-byte (OCH 3) store.  The kernels drain once per block of steps,
// right before they issue their next stream loads: the stores are then OLDER than the loads the
// wavefront next waits for, so no step ever waits for a store acknowledgement (memory operations
// retire in order; a store issued after the loads would make every wait for the loads a wait for the
// store as well).  Pixels of a group shared with the neighbouring segment (head / tail) are written
// one by one.
// ---------------------------------------------------------------------------------
#ifndef QOIMI_DRAIN_GROUP
#define QOIMI_DRAIN_GROUP 4
#endif
// RING_: pixels the LDS ring of a lane holds (power of two).  GROUP_: pixels written together - an aligned group leaves as
// GROUP_/4 back-to-back 16-byte stores of the same lane.  What the memory side makes of a lane's output stream depends on
// how close in time the pieces of a cache line arrive (tools/ubench/store_patterns.hip, MI355X, 9 GB in 64-lane wavefronts
// with 6.9 KB between the lanes' streams): 16 bytes per drain, the pieces of a line several hundred cycles apart - 1.0-1.4
// TB/s (the L2 no longer merges them); 64 bytes per lane in one burst - 3.6 TB/s; 4 lanes x 16 bytes to one line - the same.
// dec_segments_rec therefore drains groups of 16 pixels from a ring of 32.
#ifndef QOIMI_SPLAT_ALIGN_IN_RING
#define QOIMI_SPLAT_ALIGN_IN_RING 1
#endif
template <int OCH, uint32_t RING_ = 16, uint32_t GROUP_ = QOIMI_DRAIN_GROUP>
struct LaneWriter {
    static constexpr uint32_t kRing = RING_;
    static_assert((RING_ & (RING_ - 1u)) == 0u && GROUP_ % 4u == 0u && GROUP_ < RING_, "ring / group sizes");
    uint32_t row;         // LDS byte address of row[0][lane]; pixel i at row + (i & (kRing-1))*256
    uint8_t* out;
    uint32_t ppos;        // next pixel index
    uint32_t fpos;        // first pixel still in the ring

    __device__ __forceinline__ void init(uint32_t row_addr, uint8_t* image, uint32_t px_pos) {
        row = row_addr; out = image; ppos = px_pos; fpos = px_pos;
    }
    __device__ __forceinline__ uint32_t at(uint32_t i) const { return *(const lds_u32*)(row + (i & (kRing - 1u)) * 256u); }
    __device__ __forceinline__ void store_one(uint32_t i, uint32_t px) {
        if (OCH == 4) reinterpret_cast<uint32_t*>(out)[i] = px;
        else { uint8_t* d = out + (size_t)i * 3u; d[0] = (uint8_t)px; d[1] = (uint8_t)(px >> 8); d[2] = (uint8_t)(px >> 16); }
    }
    // write out every complete aligned group of kGroup pixels; a leading partial group (segment head) pixel by pixel
    static constexpr uint32_t kGroup = GROUP_;
    __device__ __forceinline__ void store4(uint32_t i, uint32_t v0, uint32_t v1, uint32_t v2, uint32_t v3) {
        if (OCH == 4) {
            *reinterpret_cast<uint4*>(out + (size_t)i * 4u) = make_uint4(v0, v1, v2, v3);
        } else {                                                      // 4 pixels -> 3 dwords of packed r,g,b
            const uint32_t a = v0 & 0xFFFFFFu, b = v1 & 0xFFFFFFu, c = v2 & 0xFFFFFFu, e = v3 & 0xFFFFFFu;
            uint32_t* d = reinterpret_cast<uint32_t*>(out + (size_t)i * 3u);
            d[0] = a | (b << 24); d[1] = (b >> 8) | (c << 16); d[2] = (c >> 16) | (e << 8);
        }
    }
    __device__ __forceinline__ void drain() {
        while ((fpos & (kGroup - 1u)) != 0u && fpos < ppos) {             // segment head / after a splat: up to the next group boundary
            if (kGroup > 4u && (fpos & 3u) == 0u && fpos + 4u <= ppos) { store4(fpos, at(fpos), at(fpos + 1u), at(fpos + 2u), at(fpos + 3u)); fpos += 4u; }
            else { store_one(fpos, at(fpos)); ++fpos; }
        }
        while (fpos + kGroup <= ppos) {
            uint32_t v[kGroup];
#pragma unroll
            for (uint32_t k = 0; k < kGroup; k += 4u) {                   // an aligned group of 4 never wraps in the ring
                const lds_u32* g = (const lds_u32*)(row + ((fpos + k) & (kRing - 1u)) * 256u);
                v[k] = g[0]; v[k + 1u] = g[64]; v[k + 2u] = g[128]; v[k + 3u] = g[192];
            }
#pragma unroll
            for (uint32_t k = 0; k < kGroup; k += 4u) store4(fpos + k, v[k], v[k + 1u], v[k + 2u], v[k + 3u]);
            fpos += kGroup;
        }
    }
    // px at ppos and at ppos + 1, the second one counted only if `two` (else it is overwritten by the next put):
    // no branch; the caller guarantees two free places
    __device__ __forceinline__ void put2(uint32_t px, bool two) {
        *(lds_u32*)(row + (ppos & (kRing - 1u)) * 256u) = px;
        *(lds_u32*)(row + ((ppos + 1u) & (kRing - 1u)) * 256u) = px;
        ppos += two ? 2u : 1u;
    }
    // the same with a count: n = 0 (nothing), 1 or 2
    __device__ __forceinline__ void put2n(uint32_t px, uint32_t n) {
        *(lds_u32*)(row + (ppos & (kRing - 1u)) * 256u) = px;
        *(lds_u32*)(row + ((ppos + 1u) & (kRing - 1u)) * 256u) = px;
        ppos += n;
    }
    __device__ __forceinline__ void put(uint32_t px) {
        if (__builtin_expect(ppos - fpos == kRing, 0)) drain();           // only long runs fill the ring between drains
        *(lds_u32*)(row + (ppos & (kRing - 1u)) * 256u) = px;
        ++ppos;
    }
    // n copies of px from ppos on, not through the ring; n is reduced to the < 4 copies that remain for put()
    __device__ __forceinline__ void splat(uint32_t px, uint32_t& n) {
#if QOIMI_SPLAT_ALIGN_IN_RING
        // up to the next multiple of four through the ring: what finish() then writes are whole 16-byte pieces (it took
        // up to three single-pixel stores for the ring's last pixels and up to three more to align the run)
        while ((ppos & 3u) != 0u && n) { put(px); --n; }
#endif
        finish();                                                         // ring out first: [fpos, ppos) stays contiguous
        while ((ppos & 3u) != 0u && n) { store_one(ppos, px); ++ppos; --n; }
        while (n >= 4u) {
            if (OCH == 4) {
                *reinterpret_cast<uint4*>(out + (size_t)ppos * 4u) = make_uint4(px, px, px, px);
            } else {
                const uint32_t a = px & 0xFFFFFFu;
                uint32_t* d = reinterpret_cast<uint32_t*>(out + (size_t)ppos * 3u);
                d[0] = a | (a << 24); d[1] = (a >> 8) | (a << 16); d[2] = (a >> 16) | (a << 8);
            }
            ppos += 4u; n -= 4u;
        }
        fpos = ppos;
    }
    __device__ __forceinline__ void finish() {                            // everything left, groups of 4 first, the rest pixel by pixel
        drain();
        while (fpos < ppos) {
            if ((fpos & 3u) == 0u && fpos + 4u <= ppos) { store4(fpos, at(fpos), at(fpos + 1u), at(fpos + 2u), at(fpos + 3u)); fpos += 4u; }
            else { store_one(fpos, at(fpos)); ++fpos; }
        }
    }
};

// ---------------------------------------------------------------------------------
// P1: parse summaries (lane = segment)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_parse(DecParams p) {
    __shared__ uint32_t s_ring[4][LaneReader::kSlots * 64];
    __shared__ LdsLut s_lut;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    build_lut(s_lut, threadIdx.x, 256u);
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    const bool have = q < p.total_segs;
    const uint32_t img = find_image(p.images, p.n_images, have ? q : 0u);
    const DecImage im = p.images[img];
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    const uint32_t base = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t end = min(base + p.seg_bytes, im.chunks_end);
    const uint8_t* stream = p.streams + im.stream_off;
    LaneReader R;
    R.init(lds_addr_of(&s_ring[wave][lane]), stream, base, im.chunks_end + kTrailerBytes);
    // Five chains (one per possible entry phase) until they stand on the same byte - a chain's next
    // position depends only on the byte it stands on, so chains that meet stay together - then ONE cursor.
    ParseState s; parse_init(s, base);
    uint32_t m = base, add = 0;
    bool merged = false;
    bool active = have && m < end;
    for (uint32_t it = 0; lanes_where(active); ++it) {
        if (R.due(it)) R.refill(m);
        if (active) {
            uint32_t w32, b5; R.peek(m, w32, b5);
            const uint32_t b1 = w32 & 0xFFu;
            if (!merged) {
                parse_step(s, m, b1);
                m = parse_front(s);
                merged = s.p0 == s.p1 && s.p1 == s.p2 && s.p2 == s.p3 && s.p3 == s.p4;
            } else {
                add += lut_pixels(s_lut.info[b1]);
                m += len_of(b1);
            }
            active = m < end;
        }
    }
    if (merged) {
        s.p0 = s.p1 = s.p2 = s.p3 = s.p4 = m;
        s.c0 += add; s.c1 += add; s.c2 += add; s.c3 += add; s.c4 += add;
    }
    if (have) { ParseRec r; parse_finish(s, base, p.seg_bytes, r); p.parse[q] = r; }
}


// ---------------------------------------------------------------------------------
// Fine-grained P1 / P2.  These two passes write only a few bytes per segment, so a segment is cut
// into 128-byte pieces, one LANE per piece, and the pieces' records are composed back into the
// segment's record inside the kernel (both records compose associatively).  A wavefront then runs
// ~100 steps instead of ~1700 and the grid has 16x the wavefronts: the lane-per-segment versions are
// bound by the serial instruction stream of their few, long-lived wavefronts (each step costs a
// wavefront ~9 cycles per instruction), not by memory or by the SIMDs.  A lane's 128 bytes (+ the 8
// bytes a chunk that starts at its end may reach, + alignment) are loaded up front into a LINEAR
// per-lane LDS buffer ([dword][lane]): no ring, no refill.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t slot_pack(const SlotRec& r) {
    return r.hc | ((uint32_t)r.h_rel << 8) | ((uint32_t)r.h_alpha << 9) | ((uint32_t)r.a_abs << 10) | ((uint32_t)r.ac << 16);
}
__device__ __forceinline__ SlotRec slot_unpack(uint32_t w) {
    SlotRec t; t.hc = w & 63u; t.h_rel = (w >> 8) & 1u; t.h_alpha = (w >> 9) & 1u; t.a_abs = (w >> 10) & 1u; t.ac = (w >> 16) & 0xFFu;
    return t;
}

#ifndef QOIMI_SYNC_RETRY
#define QOIMI_SYNC_RETRY 192
#endif
constexpr uint32_t kSyncRetryBytes = QOIMI_SYNC_RETRY;      // dec_transcode<0>: the second run-up of lanes whose chains did not meet in the first (0: none)
constexpr uint32_t kFineBytes = 128;
constexpr uint32_t kFinePieces = 10;                    // 16-byte loads per lane: 15 (alignment) + 128 + 8 <= 160
constexpr uint32_t kFineDwords = kFinePieces * 4;

struct FineBuf {
    uint32_t buf;              // LDS byte address of buf[0][lane]; dword k at buf + k*256
    uint32_t aoff;             // stream position of buffer byte 0
    __device__ __forceinline__ void init(uint32_t buf_addr, const uint8_t* stream, uint32_t pos0, uint32_t size) {
        buf = buf_addr;
        const uint8_t* p = stream + pos0;
        const uint8_t* abase = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)15);
        const uint8_t* aend = stream + size;
        aoff = pos0 - (uint32_t)(p - abase);
        uint4 v[kFinePieces];
#pragma unroll
        for (uint32_t r = 0; r < kFinePieces; ++r)
            v[r] = abase + 16u * r < aend ? load_global16(abase + 16u * r) : make_uint4(0u, 0u, 0u, 0u);
#pragma unroll
        for (uint32_t r = 0; r < kFinePieces; ++r) {
            lds_u32* q = (lds_u32*)(buf + r * 1024u);
            q[0] = v[r].x; q[64] = v[r].y; q[128] = v[r].z; q[192] = v[r].w;
        }
    }
    __device__ __forceinline__ uint32_t byte(uint32_t pos) const {
        const uint32_t rp = pos - aoff;
        const uint32_t d = *(const lds_u32*)(buf + (rp >> 2) * 256u);
        return (d >> ((rp & 3u) * 8u)) & 0xFFu;
    }
    __device__ __forceinline__ void peek(uint32_t pos, uint32_t& w32, uint32_t& b5) const {
        const uint32_t rp = pos - aoff;
        const lds_u32* q = (const lds_u32*)(buf + (rp >> 2) * 256u);
        const uint32_t d0 = q[0], d1 = q[64];
        const uint32_t sh = (rp & 3u) * 8u;
        w32 = __builtin_amdgcn_alignbit(d1, d0, sh);
        b5 = (d1 >> sh) & 0xFFu;
    }
    // the same without extracting byte 4 (needed by QOI_OP_RGBA only): hi = second dword, sh8 = 8 x the byte position
    __device__ __forceinline__ void peek_raw(uint32_t pos, uint32_t& w32, uint32_t& hi, uint32_t& sh8) const {
        const uint32_t rp = pos - aoff;
        const lds_u32* q = (const lds_u32*)(buf + (rp >> 2) * 256u);
        const uint32_t d0 = q[0];
        hi = q[64];
        sh8 = rp * 8u;
        w32 = __builtin_amdgcn_alignbit(hi, d0, sh8);                          // alignbit takes the shift modulo 32
    }
};

// The five-phase parse of the segments dec_transcode<0> could not synchronise (it flags them in sync_fail): chunk lengths
// and pixel counts for every possible entry phase; the transcoder (MODE 1) then walks them from the phase S1 finds.
__global__ __launch_bounds__(256) void dec_parse_fine(DecParams p) {
    __shared__ uint32_t s_buf[4][kFineDwords * 64];
    __shared__ uint32_t s_rec[4][64][6];          // per lane: exit map, pixels[5]
    __shared__ uint32_t s_res[4][64][2];          // per (group, entry phase): exit phase, pixels
    __shared__ LdsLut s_lut;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    if (*p.sync_fails == 0u) return;                       // nothing to parse (dec_transcode synchronised every segment)
    const uint32_t G = p.fine_per_seg, gs = p.fine_shift;
    const uint32_t F = blockIdx.x * 256u + threadIdx.x;
    const uint32_t q = F >> gs, sub = F & (G - 1u);
    bool have = q < p.total_segs;
    have = have && p.sync_fail[have ? q : 0u] != 0u;      // only the segments dec_transcode could not synchronise
    // (a workgroup without one returns before it builds its table: a call with a handful of flagged segments was 11 000 workgroups that
    // each built one - 65 of the mixed directory's 3700 us, profiles/r06_s33_mixed_timeline.txt)
    if (!__syncthreads_or(have ? 1 : 0)) return;
    build_lut(s_lut, threadIdx.x, 256u);
    if (!lanes_where(have)) return;
    const uint32_t img = find_image(p.images, p.n_images, have ? q : 0u);
    const DecImage im = p.images[img];
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    const uint32_t cbase = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t cend = min(cbase + p.seg_bytes, im.chunks_end);
    const uint32_t base = cbase + sub * kFineBytes;
    const uint32_t end = min(base + kFineBytes, cend);
    FineBuf R;
    R.init(lds_addr_of(&s_buf[wave][lane]), p.streams + im.stream_off, min(base, im.chunks_end), im.chunks_end + kTrailerBytes);
    // five chains (one per possible entry phase) until they stand on the same byte, then ONE cursor
    ParseState s; parse_init(s, base);
    uint32_t m = base;
    bool merged = false;
    bool active = have && m < end;
    while (lanes_where(active && !merged)) {
        if (active && !merged) {
            parse_step(s, m, R.byte(m));
            m = parse_front(s);
            merged = s.p0 == s.p1 && s.p1 == s.p2 && s.p2 == s.p3 && s.p3 == s.p4;
            active = m < end;
        }
    }
    // From the byte where the chains met, the chunk sequence no longer depends on the entry phase.
    uint32_t add = 0;
    // one chunk per step; the next chunk's bytes and its table word are asked for before this chunk's arithmetic
    // No lane is masked off inside the loop: a lane that is through with its piece runs a null chunk (length 0, no
    // pixels, slot shift 0) - with the exec mask untouched the loop has no merge copies; two steps per iteration with
    // the chunk registers swapping roles, so that the next chunk's words need no copy at the loop's end either.
    const uint32_t end_b = active ? end : 0u;                              // lanes that take no part: m < 0 never holds
    uint32_t wa, ha, sa; R.peek_raw(min(m, end), wa, ha, sa);
    uint32_t ia = s_lut.info[wa & 0xFFu];
    auto step = [&](uint32_t w32, uint32_t hiw, uint32_t sh8, uint32_t info, uint32_t& nw32, uint32_t& nhi, uint32_t& nsh, uint32_t& ninfo) {
        const bool on = m < end_b;
        const uint32_t c_info = on ? info : 0u;
        m += lut_len(c_info);                                              // this kernel is VALU-bound with twelve wavefronts per CU:
                                                                           // the table's length (2 ops) beats the arithmetic one (7)
        R.peek_raw(m, nw32, nhi, nsh);                                     // stays inside the buffer: m <= end + 4
        ninfo = s_lut.info[nw32 & 0xFFu];
        add += lut_pixels(c_info);
        (void)w32; (void)hiw; (void)sh8;
    };
    while (lanes_where(m < end_b)) {
        uint32_t wb, hb, sb, ib;
        step(wa, ha, sa, ia, wb, hb, sb, ib);
        step(wb, hb, sb, ib, wa, ha, sa, ia);
    }
    if (merged) {
        s.p0 = s.p1 = s.p2 = s.p3 = s.p4 = m;
        s.c0 += add; s.c1 += add; s.c2 += add; s.c3 += add; s.c4 += add;
    }
    ParseRec r; parse_finish(s, base, kFineBytes, r);
    if (G == 1u) { if (have) p.parse[q] = r; return; }     // 128-byte segments: the piece is the segment
    // compose the G (8..64) pieces of every segment: lane `sub` = e < 5 walks the pieces for entry phase e
    s_rec[wave][lane][0] = r.exit_phase;
#pragma unroll
    for (int k = 0; k < 5; ++k) s_rec[wave][lane][1 + k] = r.pixels[k];
    __builtin_amdgcn_wave_barrier();
    const uint32_t g0 = lane - sub;                      // first lane of this segment's group
    if (G < 8u) {                                        // two or four pieces (256- and 512-byte segments): fewer lanes than entry phases - the segment's first lane walks all five
        if (have && sub == 0u) {
            ParseRec o; o.exit_phase = 0;
#pragma unroll
            for (uint32_t e = 0; e < 5u; ++e) {
                uint32_t ph = e, tot = 0;
                for (uint32_t k = 0; k < G; ++k) {
                    tot += s_rec[wave][g0 + k][1u + ph];
                    ph = (s_rec[wave][g0 + k][0] >> (3u * ph)) & 7u;
                }
                o.exit_phase |= ph << (3u * e);
                o.pixels[e] = tot;
            }
            p.parse[q] = o;
        }
        return;
    }
    if (sub < 5u) {
        uint32_t ph = sub, tot = 0;
        for (uint32_t k = 0; k < G; ++k) {
            tot += s_rec[wave][g0 + k][1u + ph];
            ph = (s_rec[wave][g0 + k][0] >> (3u * ph)) & 7u;
        }
        s_res[wave][lane][0] = ph; s_res[wave][lane][1] = tot;
    }
    __builtin_amdgcn_wave_barrier();
    if (have && sub == 0u) {
        ParseRec o; o.exit_phase = 0;
#pragma unroll
        for (int e = 0; e < 5; ++e) {
            o.exit_phase |= s_res[wave][lane + e][0] << (3 * e);
            o.pixels[e] = s_res[wave][lane + e][1];
        }
        p.parse[q] = o;
    }
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
// The records of S1 and S2 compose associatively, so 64 of them (one per lane) are chained with a log-step scan
// over the wavefront - six rounds - instead of 64 dependent steps on broadcast lanes (22 us per kernel on a
// single frame, where these chains are a good part of the whole decode).
// Phase maps: 3 bits per entry phase; (a then b)[k] = b[a[k]].
constexpr uint32_t kMapIdentity = 0u | (1u << 3) | (2u << 6) | (3u << 9) | (4u << 12);
__device__ __forceinline__ uint32_t map_compose(uint32_t a, uint32_t b) {
    uint32_t r = 0;
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) r |= ((b >> (3u * ((a >> (3u * k)) & 7u))) & 7u) << (3u * k);
    return r;
}
__device__ __forceinline__ uint32_t sat_add(uint32_t a, uint32_t b) { const uint32_t s = a + b; return s < a ? 0xFFFFFFFFu : s; }
// inclusive scans over the 64 lanes
__device__ __forceinline__ uint32_t wave_scan_map(uint32_t m, uint32_t lane) {
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t up = gather_lane(m, lane - d); if (lane >= d) m = map_compose(up, m); }
    return m;
}
__device__ __forceinline__ uint32_t wave_scan_sat(uint32_t v, uint32_t lane) {
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) { const uint32_t up = gather_lane(v, lane - d); if (lane >= d) v = sat_add(up, v); }
    return v;
}
// record of 64 consecutive records (lane l holds record l; lanes past the end hold the identity): the composed map
// and, per entry phase, the pixels along that phase's path (saturating: far above any legal pixel count anyway)
__device__ __forceinline__ ParseRec parse_fold64(const ParseRec& r, uint32_t lane) {
    const uint32_t incl = wave_scan_map(r.exit_phase, lane);
    const uint32_t excl = from_lane_below(incl, kMapIdentity);
    ParseRec g;
    g.exit_phase = read_lane(incl, 63);
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) {
        const uint32_t ph = (excl >> (3u * k)) & 7u;
        g.pixels[k] = read_lane(wave_scan_sat(sel5(ph, r.pixels[0], r.pixels[1], r.pixels[2], r.pixels[3], r.pixels[4]), lane), 63);
    }
    return g;
}
// entry (phase, pixels before) of every one of the 64 records for a concrete entry phase; total: the pixels of all 64
__device__ __forceinline__ void parse_sweep64(const ParseRec& r, uint32_t lane, uint32_t phase0, uint32_t& my_phase, uint32_t& my_before,
                                              uint32_t& exit_phase, uint32_t& total) {
    const uint32_t incl = wave_scan_map(r.exit_phase, lane);
    const uint32_t excl = from_lane_below(incl, kMapIdentity);
    my_phase = (excl >> (3u * phase0)) & 7u;
    const uint32_t sums = wave_scan_sat(sel5(my_phase, r.pixels[0], r.pixels[1], r.pixels[2], r.pixels[3], r.pixels[4]), lane);
    my_before = from_lane_below(sums, 0u);
    total = read_lane(sums, 63);
    exit_phase = (read_lane(incl, 63) >> (3u * phase0)) & 7u;
}
__device__ __forceinline__ ParseRec parse_identity() {
    ParseRec r; r.exit_phase = kMapIdentity;
#pragma unroll
    for (int k = 0; k < 5; ++k) r.pixels[k] = 0;
    return r;
}

__global__ __launch_bounds__(64) void dec_chain_parse_l1(DecParams p) {
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 >= im.nseg) return;                              // a padding group between two images (DecParams::fused)
    const uint32_t cnt = min(kGrp, im.nseg - j0);
    ParseRec r = parse_identity();
    if (lane < cnt) r = p.parse[im.seg_base + j0 + lane];
    const ParseRec g = parse_fold64(r, lane);
    if (lane == 0) p.grp_parse[G] = g;
}

// The l2 kernels chain the group records of ONE image.  A single 4K frame has hundreds of groups, a 16384 x 16384
// image thousands, and a lone wavefront walking them one after the other was most of the decode time of such
// calls.  One block of kL2Waves wavefronts per image instead: every wavefront composes the records of its
// contiguous share of the groups (A), wavefront 0 chains the kL2Waves shares (B), every wavefront sweeps its share
// from the entry value it was given (C): serial depth 2 * ngrp / kL2Waves + kL2Waves instead of ngrp.
constexpr uint32_t kL2Waves = 16;

__global__ __launch_bounds__(64 * kL2Waves) void dec_chain_parse_l2(DecParams p) {
    __shared__ uint32_t s_rec[kL2Waves][6];      // share record: exit-phase map, pixels[5]
    __shared__ uint32_t s_in[kL2Waves][2];       // share entry: phase, pixel offset
    const uint32_t img = blockIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const DecImage im = p.images[img];
    if (wave == kL2Waves - 1u && im.nseg != 0u) {          // the decoder's start state (qoi.h:533-537) at the image's first segment
        p.entry[(size_t)im.seg_base * 65u + lane] = 0u;
        if (lane == 0) p.entry[(size_t)im.seg_base * 65u + 64u] = kInitPx;
    }
    const uint32_t per = (im.ngrp + kL2Waves - 1u) / kL2Waves;
    const uint32_t lo = min(wave * per, im.ngrp), hi = min(lo + per, im.ngrp);
    {   // A: record of the share, 64 groups per fold
        ParseRec S = parse_identity();
        for (uint32_t g0 = lo; g0 < hi; g0 += 64u) {
            ParseRec r = parse_identity();
            if (g0 + lane < hi) r = p.grp_parse[im.grp_base + g0 + lane];
            const ParseRec b = parse_fold64(r, lane);
            ParseRec n;
            n.exit_phase = map_compose(S.exit_phase, b.exit_phase);
#pragma unroll
            for (uint32_t k = 0; k < 5u; ++k)
                n.pixels[k] = sat_add(S.pixels[k], sel5((S.exit_phase >> (3u * k)) & 7u, b.pixels[0], b.pixels[1], b.pixels[2], b.pixels[3], b.pixels[4]));
            S = n;
        }
        if (lane == 0) {
            s_rec[wave][0] = S.exit_phase;
#pragma unroll
            for (uint32_t k = 0; k < 5u; ++k) s_rec[wave][1u + k] = S.pixels[k];
        }
    }
    __syncthreads();
    // B
    if (wave == 0 && lane == 0) {
        uint32_t phase = 0; u64 off = 0;
        for (uint32_t c = 0; c < kL2Waves; ++c) {
            s_in[c][0] = phase; s_in[c][1] = (uint32_t)off;
            off = min(off + s_rec[c][1u + phase], (u64)im.npx);
            phase = (s_rec[c][0] >> (3u * phase)) & 7u;
        }
        p.images[img].total_px = (uint32_t)off;
        p.images[img].n_active = 0;              // raised by l3
        p.images[img].start_seg = 0;
        p.first_bad[img] = 0xFFFFFFFFu;
    }
    __syncthreads();
    // C
    uint32_t phase = s_in[wave][0]; u64 off = s_in[wave][1];
    for (uint32_t g0 = lo; g0 < hi; g0 += 64u) {
        ParseRec r = parse_identity();
        if (g0 + lane < hi) r = p.grp_parse[im.grp_base + g0 + lane];
        uint32_t my_phase, my_before, exit_phase, total;
        parse_sweep64(r, lane, phase, my_phase, my_before, exit_phase, total);
        if (g0 + lane < hi) {
            p.grp_phase[im.grp_base + g0 + lane] = (uint8_t)my_phase;
            p.grp_off[im.grp_base + g0 + lane] = (uint32_t)min(off + my_before, (u64)im.npx);
        }
        off = min(off + total, (u64)im.npx);
        phase = exit_phase;
    }
}

__global__ __launch_bounds__(64) void dec_chain_parse_l3(DecParams p) {
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 >= im.nseg) return;                              // (a padding group)
    const uint32_t cnt = min(kGrp, im.nseg - j0);
    ParseRec r = parse_identity();
    if (lane < cnt) r = p.parse[im.seg_base + j0 + lane];
    const uint32_t phase0 = p.grp_phase[G]; const u64 off0 = p.grp_off[G];
    uint32_t my_phase, my_before, exit_phase, total;
    parse_sweep64(r, lane, phase0, my_phase, my_before, exit_phase, total);
    const uint32_t my_off = (uint32_t)min(off0 + my_before, (u64)im.npx);
    if (lane < cnt) { p.entry_phase[im.seg_base + j0 + lane] = (uint8_t)my_phase; p.px_off[im.seg_base + j0 + lane] = my_off; }
    // n_active: segments that start before the pixel limit.  Only the group that holds the last such segment
    // reports (its successor starts at the limit, or it is the image's last group): one atomic per image, not one per
    // group on the same word.
    const u64 act = lanes_where(lane < cnt && my_off < im.npx);
    const bool boundary = j0 + cnt == im.nseg || off0 + total >= (u64)im.npx;
    if (lane == 0 && act && boundary) atomicMax(&p.images[img].n_active, j0 + 64u - (uint32_t)__builtin_clzll(act));
}

// S2 with the same log-step scans.  Lane l holds the packed transfer of record l (identity outside the range).
constexpr uint32_t kSlotIdentity = 1u << 8;                    // slot_pack({0, 1, 0, 0, 0})
__device__ __forceinline__ uint32_t wave_scan_slots(uint32_t w, uint32_t lane) {
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up = gather_lane(w, lane - d);
        if (lane >= d) w = slot_pack(slot_compose(slot_unpack(up), slot_unpack(w)));
    }
    return w;
}

// S2 l1: compose the transfers of the group's segments that are still to be decoded
__global__ __launch_bounds__(64) void dec_chain_slots_l1(DecParams p) {
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 + kGrp <= im.start_seg || j0 >= im.n_active) return;
    const uint32_t lo = max(j0, im.start_seg), hi = min(j0 + kGrp, im.n_active);
    uint32_t mine = kSlotIdentity;
    if (j0 + lane >= lo && j0 + lane < hi) mine = slot_pack(p.slot_rec[im.seg_base + j0 + lane]);
    const uint32_t incl = wave_scan_slots(mine, lane);
    if (lane == 63u) p.grp_slot[G] = slot_unpack(incl);
}

// S2 l2: chain the groups of one image from the group holding start_seg; the start value is the
// hash/alpha of start_seg's concrete entry pixel.  Shares as in dec_chain_parse_l2.
__global__ __launch_bounds__(64 * kL2Waves) void dec_chain_slots_l2(DecParams p) {
    __shared__ uint32_t s_rec[kL2Waves];         // packed transfer of the share
    __shared__ uint32_t s_in[kL2Waves][2];       // share entry: slot, alpha
    const uint32_t img = blockIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const DecImage im = p.images[img];
    if (im.start_seg >= im.n_active) return;
    const uint32_t gfirst = im.start_seg / kGrp, gend = (im.n_active + kGrp - 1u) / kGrp;
    const uint32_t per = (gend - gfirst + kL2Waves - 1u) / kL2Waves;
    const uint32_t lo = min(gfirst + wave * per, gend), hi = min(lo + per, gend);
    {   // A
        SlotRec acc = {0, 1, 0, 0, 0};
        for (uint32_t g0 = lo; g0 < hi; g0 += 64u) {
            uint32_t mine = kSlotIdentity;
            if (g0 + lane < hi) mine = slot_pack(p.grp_slot[im.grp_base + g0 + lane]);
            acc = slot_compose(acc, slot_unpack(read_lane(wave_scan_slots(mine, lane), 63)));
        }
        if (lane == 0) s_rec[wave] = slot_pack(acc);
    }
    __syncthreads();
    if (wave == 0 && lane == 0) {   // B
        const uint32_t px0 = p.entry[(size_t)(im.seg_base + im.start_seg) * 65u + 64u];
        uint32_t slot = hash_px(px0), alpha = px0 >> 24;
        for (uint32_t c = 0; c < kL2Waves; ++c) {
            s_in[c][0] = slot; s_in[c][1] = alpha;
            slot_apply(slot_unpack(s_rec[c]), slot, alpha);
        }
    }
    __syncthreads();
    // C
    uint32_t slot = s_in[wave][0], alpha = s_in[wave][1];
    for (uint32_t g0 = lo; g0 < hi; g0 += 64u) {
        uint32_t mine = kSlotIdentity;
        if (g0 + lane < hi) mine = slot_pack(p.grp_slot[im.grp_base + g0 + lane]);
        const uint32_t incl = wave_scan_slots(mine, lane);
        uint32_t my_slot = slot, my_alpha = alpha;
        slot_apply(slot_unpack(from_lane_below(incl, kSlotIdentity)), my_slot, my_alpha);
        if (g0 + lane < hi) { p.grp_slot_in[im.grp_base + g0 + lane] = (uint8_t)my_slot; p.grp_alpha_in[im.grp_base + g0 + lane] = (uint8_t)my_alpha; }
        slot_apply(slot_unpack(read_lane(incl, 63)), slot, alpha);
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
    uint32_t mine = kSlotIdentity;
    if (j0 + lane >= lo && j0 + lane < hi) mine = slot_pack(p.slot_rec[im.seg_base + j0 + lane]);
    const uint32_t incl = wave_scan_slots(mine, lane);
    uint32_t my_slot = p.grp_slot_in[G], my_alpha = p.grp_alpha_in[G];
    slot_apply(slot_unpack(from_lane_below(incl, kSlotIdentity)), my_slot, my_alpha);
    if (j0 + lane >= lo && j0 + lane < hi) { p.slot_in[im.seg_base + j0 + lane] = (uint8_t)my_slot; p.alpha_in[im.seg_base + j0 + lane] = (uint8_t)my_alpha; }
}

// ---------------------------------------------------------------------------------
// Symbolic words of P3 (dec_summarize_rec) and the private colour table of P4 (dec_segments_rec)
// ---------------------------------------------------------------------------------
// The symbolic table is kept as two LDS arrays, constants (u32) and a one-byte code for source and absolute mask:
// 20 KiB per wavefront instead of 32, which lets a sixth wavefront onto the CU (this kernel is bound by how many
// lanes are resident, see dec_segments).  Only three masks occur - nothing absolute, r,g,b absolute (QOI_OP_RGB
// keeps the alpha), everything absolute (QOI_OP_RGBA; the source no longer matters) - so the code is
//   source (0..64)          nothing absolute
//   65 + source             r,g,b absolute
//   255                     all absolute
// Same step as symf_step (qoi_decode_core.h), with the SDWA byte adds and with the LDS round trips of a step in
// flight together.
//
// REFINE (rounds after a failed exit-state check): the entry states the previous round computed serve as
// hints - slot and alpha at segment entry come from the hinted entry pixel (no P2/S2 in these rounds), and
// an INDEX chunk that names a table entry whose alpha still refers to the entry state takes the hinted
// alpha of that entry word.  Streams whose alpha changes through the colour table (UI content with
// several alpha levels) mis-speculate the slot of a later QOI_OP_RGB in almost every segment in round 1 and
// verify in 2-5 rounds with the hints (one segment per round without them).
constexpr uint32_t kSymCodeRgb = 65u, kSymCodeAbs = 255u;
__device__ __forceinline__ uint32_t sym_code_expand(uint32_t code) {       // -> source | absmask << 8 (sym_t bits 32..43)
    return code == kSymCodeAbs ? (15u << 8) : (code >= kSymCodeRgb ? ((code - kSymCodeRgb) | (7u << 8)) : code);
}


// a lane's private 64-entry colour table in LDS, laid out [slot][lane]: a lane always hits its own bank
struct LdsTab32 {
    uint32_t* col;
    __device__ __forceinline__ uint32_t get(uint32_t k) const { return col[k * 64u]; }
    __device__ __forceinline__ void set(uint32_t k, uint32_t v) { col[k * 64u] = v; }
};

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
// The PIXEL word of a record is the same in every lane (all 64 load rec[65 k + 64]) and so is the running pixel word: its step is
// done once, on the scalar side - the word made known as wave-uniform (readfirstlane), its source read with v_readlane (a scalar lane
// index: no trip through the LDS crossbar as ds_bpermute takes) and composed in SGPRs.  A step of the S3 chains was two table gathers
// and two pixel gathers + twice the arithmetic in every lane (0.17 us per step, 188 dependent steps on a lone 4K frame); the pixel
// half now rides along with the table half.
__device__ __forceinline__ sym_t uniform_sym(sym_t v) {
    return (sym_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)v) | ((sym_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(v >> 32)) << 32);
}
__device__ __forceinline__ sym_t read_sym_lane(sym_t tabv, sym_t pxv_u, uint32_t src_u) {       // src_u: wave-uniform
    const uint32_t lo = read_lane_dyn((uint32_t)tabv, src_u & 63u), hi = read_lane_dyn((uint32_t)(tabv >> 32), src_u & 63u);
    return src_u == 64u ? pxv_u : ((sym_t)lo | ((sym_t)hi << 32));
}

// The two loops of the S3 kernels.  A record is 65 symbolic words (lane k holds word k, the pixel word is read by
// every lane); a step needs the state the previous step left, so the loop is a dependent chain of gathers - but the
// RECORDS do not depend on it: they are fetched kChainBatch steps ahead, two batches in registers (with one record
// of look-ahead a step cost a memory round trip, 0.4 us; the chain itself is a third of that).
// (Round 6: batches of 32 - a group's 64 records all asked for when its chain starts - made these kernels SLOWER (l1 14.8 -> 17.1 us on
// a lone 4K frame, the 1024-thread l2 kernels spill): a step is its own dependent chain of a gather and a dozen vector instructions,
// 0.17 us, not a wait for records.  The pixel words of a batch travel in ONE load - lane i holds record i's - and are read with
// v_readlane at the step.)
constexpr int kChainBatch = 8;
struct SymBatch { sym_t tab[kChainBatch]; sym_t px; };
// records first .. first+kChainBatch-1 of rec[0..count), indices clamped to the last record: always kChainBatch
// loads, so the compiler can count them (loads under a branch made every wait a wait for ALL outstanding loads,
// the prefetched batch included)
__device__ __forceinline__ void sym_batch_load(SymBatch& b, const sym_t* __restrict__ rec, uint32_t first, uint32_t count, uint32_t lane) {
    b.px = rec[(size_t)min(first + lane, count - 1u) * 65u + 64u];
#pragma unroll
    for (int i = 0; i < kChainBatch; ++i) {
        const uint32_t k = min(first + (uint32_t)i, count - 1u);
        b.tab[i] = rec[(size_t)k * 65u + lane];
    }
}
__device__ __forceinline__ sym_t batch_px(const SymBatch& b, int i) {           // record i's pixel word, wave-uniform
    return (sym_t)read_lane((uint32_t)b.px, i) | ((sym_t)read_lane((uint32_t)(b.px >> 32), i) << 32);
}
// one step of a symbolic chain: P <- c o P.  c_px_u, P_px and the result's pixel word are wave-uniform.
__device__ __forceinline__ void sym_chain_step(sym_t c_tab, sym_t c_px_u, sym_t& P_tab, sym_t& P_px) {
    const sym_t n_tab = sym_compose(c_tab, gather_sym(P_tab, P_px, sym_src(c_tab)));
    const sym_t n_px = sym_compose(c_px_u, read_sym_lane(P_tab, P_px, sym_src(c_px_u)));
    P_tab = n_tab; P_px = n_px;
}
// one step of a concrete sweep: (tabv, pxv) <- c applied to (tabv, pxv).  c_px_u and pxv are wave-uniform.
__device__ __forceinline__ void sym_sweep_step(sym_t c_tab, sym_t c_px_u, uint32_t& tabv, uint32_t& pxv) {
    const uint32_t src_t = sym_src(c_tab), src_p = sym_src(c_px_u);
    const uint32_t g_t = gather_lane(tabv, src_t & 63u), g_p = read_lane_dyn(tabv, src_p & 63u);
    const uint32_t ntab = sym_eval(c_tab, src_t == 64u ? pxv : g_t);
    const uint32_t npx = sym_eval(c_px_u, src_p == 64u ? pxv : g_p);
    tabv = ntab; pxv = npx;
}
// P <- rec[count-1] o ... o rec[0] o P  (symbolic composition)
__device__ __forceinline__ void sym_compose_range(const sym_t* __restrict__ rec, uint32_t count, uint32_t lane, sym_t& P_tab, sym_t& P_px) {
    if (count == 0u) return;
    SymBatch cur, nxt;
    sym_batch_load(cur, rec, 0u, count, lane);
    for (uint32_t g = 0; g < count; g += kChainBatch) {
        const uint32_t n = min(count - g, (uint32_t)kChainBatch);
        sym_batch_load(nxt, rec, g + kChainBatch, count, lane);
#pragma unroll
        for (int i = 0; i < kChainBatch; ++i) {
            if ((uint32_t)i < n) sym_chain_step(cur.tab[i], batch_px(cur, i), P_tab, P_px);
        }
        cur = nxt;
    }
}
// concrete state pushed through rec[0..count); the state BEFORE record i is written to out + i*65 for i >= first_out
// TRACK (refinement passes, DecParams::conv): `changed` is raised where a state written differs from the one it replaces in what
// dec_summarize_rec<true> reads from it - the alpha byte of every word, the slot of the entry pixel.  The old words are fetched a batch ahead
// of their stores, as the records are.
template <bool TRACK = false>
__device__ __forceinline__ void sym_sweep_range(const sym_t* __restrict__ rec, uint32_t count, uint32_t lane, uint32_t& tabv, uint32_t& pxv,
                                                uint32_t* out, uint32_t first_out, bool* changed = nullptr) {
    if (count == 0u) return;
    pxv = (uint32_t)__builtin_amdgcn_readfirstlane((int)pxv);               // (the same in every lane: say so, the pixel half of a step is scalar work)
    SymBatch cur, nxt;
    uint32_t old_c[kChainBatch], old_n[kChainBatch], oldpx_c = 0u, oldpx_n = 0u;
    auto old_load = [&](uint32_t (&o)[kChainBatch], uint32_t& opx, uint32_t first) {
        opx = out[(size_t)min(first + lane, count - 1u) * 65u + 64u];
#pragma unroll
        for (int i = 0; i < kChainBatch; ++i) o[i] = out[(size_t)min(first + (uint32_t)i, count - 1u) * 65u + lane];
    };
    sym_batch_load(cur, rec, 0u, count, lane);
    if (TRACK) old_load(old_c, oldpx_c, 0u);
    uint32_t diff = 0u;
    for (uint32_t g = 0; g < count; g += kChainBatch) {
        const uint32_t n = min(count - g, (uint32_t)kChainBatch);
        sym_batch_load(nxt, rec, g + kChainBatch, count, lane);
        if (TRACK) old_load(old_n, oldpx_n, g + kChainBatch);
#pragma unroll
        for (int i = 0; i < kChainBatch; ++i) {
            if ((uint32_t)i < n) {
                if (g + (uint32_t)i >= first_out) {
                    out[(size_t)(g + i) * 65u + lane] = tabv;
                    if (lane == 0) out[(size_t)(g + i) * 65u + 64u] = pxv;
                    if (TRACK) {
                        const uint32_t opx = read_lane(oldpx_c, i);
                        diff |= (old_c[i] ^ tabv) >> 24;
                        diff |= ((opx ^ pxv) >> 24) | (hash_px(opx) ^ hash_px(pxv));
                    }
                }
                sym_sweep_step(cur.tab[i], batch_px(cur, i), tabv, pxv);
            }
        }
        cur = nxt;
        if (TRACK) {
#pragma unroll
            for (int i = 0; i < kChainBatch; ++i) old_c[i] = old_n[i];
            oldpx_c = oldpx_n;
        }
    }
    if (TRACK) *changed = lanes_where(diff != 0u) != 0ull;
}
// refinement passes: what the kernels of a pass ask first (see DecParams::conv)
__device__ __forceinline__ bool refine_pass_idle(const DecParams& p) {
    return p.conv_pass > 1u && p.conv[p.conv_pass - 2u] == 0u;
}
__device__ __forceinline__ void refine_pass_note(const DecParams& p, bool changed, uint32_t lane) {
    if (changed && lane == 0u) atomicAdd(&p.conv[p.conv_pass - 1u], 1u);
}

__global__ __launch_bounds__(64) void dec_chain_state_l1(DecParams p) {
    if (refine_pass_idle(p)) return;
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 + kGrp <= im.start_seg || j0 >= im.n_active) return;
    const uint32_t lo = max(j0, im.start_seg), hi = min(j0 + kGrp, im.n_active);
    sym_t P_tab = sym_make(0u, lane, 0u), P_px = sym_make(0u, 64u, 0u);          // identity
    sym_compose_range(p.summary + (size_t)(im.seg_base + lo) * 65u, hi - lo, lane, P_tab, P_px);
    p.grp_summary[(size_t)G * 65u + lane] = P_tab;
    if (lane == 0) p.grp_summary[(size_t)G * 65u + 64u] = P_px;
}

__global__ __launch_bounds__(64 * kL2Waves) void dec_chain_state_l2(DecParams p) {
    if (refine_pass_idle(p)) return;
    __shared__ sym_t s_sum[kL2Waves][65];        // symbolic summary of every share
    __shared__ uint32_t s_ent[kL2Waves][65];     // concrete state at every share's entry
    const uint32_t img = blockIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const DecImage im = p.images[img];
    if (im.start_seg >= im.n_active) return;
    const uint32_t gfirst = im.start_seg / kGrp, gend = (im.n_active + kGrp - 1u) / kGrp;
    const uint32_t per = (gend - gfirst + kL2Waves - 1u) / kL2Waves;
    const uint32_t lo = min(gfirst + wave * per, gend), hi = min(lo + per, gend);
    const sym_t* __restrict__ gs = p.grp_summary + (size_t)(im.grp_base + lo) * 65u;
    {   // A: compose the share's group summaries (as dec_chain_state_l1 composes segment summaries)
        sym_t P_tab = sym_make(0u, lane, 0u), P_px = sym_make(0u, 64u, 0u);          // identity
        sym_compose_range(gs, hi - lo, lane, P_tab, P_px);
        s_sum[wave][lane] = P_tab;
        if (lane == 0) s_sum[wave][64] = P_px;
    }
    __syncthreads();
    if (wave == 0) {   // B: concrete entry state of every share
        const size_t q0 = (size_t)im.seg_base + im.start_seg;
        uint32_t tabv = p.entry[q0 * 65u + lane];        // concrete entry state of start_seg is given
        uint32_t pxv = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.entry[q0 * 65u + 64u]);
        for (uint32_t c = 0; c < kL2Waves; ++c) {
            s_ent[c][lane] = tabv;
            if (lane == 0) s_ent[c][64] = pxv;
            sym_sweep_step(s_sum[c][lane], uniform_sym(s_sum[c][64]), tabv, pxv);
        }
    }
    __syncthreads();
    // C: sweep the share
    uint32_t tabv = s_ent[wave][lane], pxv = s_ent[wave][64];
    sym_sweep_range(gs, hi - lo, lane, tabv, pxv, p.grp_entry + (size_t)(im.grp_base + lo) * 65u, 0u);
}

// The same level for a call of one to four large images (one 4K frame at 128-byte segments: 1250 groups): l2_wgs workgroups
// per image, kL2Waves shares each.  A workgroup composes its shares into ONE symbolic summary, publishes it (tag of the
// launch in its flag word), applies the summaries of the workgroups in front of it to the image's concrete start state and goes
// on as dec_chain_state_l2 does: serial depth 2 * ngrp / (16 * wgs) + 32 + wgs instead of 2 * ngrp / 16 + 16 (45 -> 18 us for
// the 4K frame).  A workgroup's place in its image is the order in which the workgroups STARTED (a ticket), so the ones it waits
// for are running whatever order the dispatcher chose.
__global__ __launch_bounds__(64 * kL2Waves) void dec_chain_state_l2m(DecParams p, uint32_t tag) {
    if (refine_pass_idle(p)) return;                       // (before the ticket: a launch takes all of its tickets or none)
    __shared__ sym_t s_sum[kL2Waves + 1][65];    // symbolic summary of every share; [kL2Waves]: scratch of the hand-over
    __shared__ uint32_t s_ent[kL2Waves][65];     // concrete state at every share's entry
    __shared__ uint32_t s_k;
    const uint32_t W = p.l2_wgs;
    const uint32_t img = blockIdx.x / W, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x == 0) s_k = atomicAdd(&p.l2_ticket[img], 1u) & (W - 1u);      // every launch takes W tickets per image
    __syncthreads();
    const uint32_t k = s_k;
    const DecImage im = p.images[img];
    if (im.start_seg >= im.n_active) return;                                      // (all W workgroups of the image)
    const uint32_t gfirst = im.start_seg / kGrp, gend = (im.n_active + kGrp - 1u) / kGrp;
    const uint32_t per = (gend - gfirst + W * kL2Waves - 1u) / (W * kL2Waves);
    const uint32_t lo = min(gfirst + (k * kL2Waves + wave) * per, gend), hi = min(lo + per, gend);
    const sym_t* __restrict__ gs = p.grp_summary + (size_t)(im.grp_base + lo) * 65u;
    {   // A: compose the share's group summaries
        sym_t P_tab = sym_make(0u, lane, 0u), P_px = sym_make(0u, 64u, 0u);          // identity
        sym_compose_range(gs, hi - lo, lane, P_tab, P_px);
        s_sum[wave][lane] = P_tab;
        if (lane == 0) s_sum[wave][64] = P_px;
    }
    __syncthreads();
    if (wave == 0) {
        {   // B1: the workgroup's summary, for the workgroups behind it
            sym_t P_tab = sym_make(0u, lane, 0u), P_px = sym_make(0u, 64u, 0u);
            sym_compose_range(&s_sum[0][0], kL2Waves, lane, P_tab, P_px);
            sym_t* out = p.l2_sum + (size_t)(img * W + k) * 65u;
            out[lane] = P_tab;
            if (lane == 0) out[64] = P_px;
            __threadfence();
            if (lane == 0) __hip_atomic_store(&p.l2_flag[img * W + k], tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        // B2: concrete state at this workgroup's entry, then at every share's entry
        const size_t q0 = (size_t)im.seg_base + im.start_seg;
        uint32_t tabv = p.entry[q0 * 65u + lane];        // concrete entry state of start_seg is given
        uint32_t pxv = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.entry[q0 * 65u + 64u]);
        for (uint32_t c = 0; c < k; ++c) {
            while (__hip_atomic_load(&p.l2_flag[img * W + c], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != tag) __builtin_amdgcn_s_sleep(2);
            const sym_t* in = p.l2_sum + (size_t)(img * W + c) * 65u;
            const sym_t c_tab = __hip_atomic_load(&in[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const sym_t c_px = __hip_atomic_load(&in[64], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            sym_sweep_step(c_tab, uniform_sym(c_px), tabv, pxv);
        }
        for (uint32_t c = 0; c < kL2Waves; ++c) {
            s_ent[c][lane] = tabv;
            if (lane == 0) s_ent[c][64] = pxv;
            sym_sweep_step(s_sum[c][lane], uniform_sym(s_sum[c][64]), tabv, pxv);
        }
    }
    __syncthreads();
    // C: sweep the share
    uint32_t tabv = s_ent[wave][lane], pxv = s_ent[wave][64];
    sym_sweep_range(gs, hi - lo, lane, tabv, pxv, p.grp_entry + (size_t)(im.grp_base + lo) * 65u, 0u);
}

__global__ __launch_bounds__(64) void dec_chain_state_l3(DecParams p) {
    if (refine_pass_idle(p)) return;
    const uint32_t G = blockIdx.x, lane = lane_id();
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 + kGrp <= im.start_seg || j0 >= im.n_active) return;
    const uint32_t lo = max(j0, im.start_seg), hi = min(j0 + kGrp, im.n_active);
    uint32_t tabv = p.grp_entry[(size_t)G * 65u + lane], pxv = p.grp_entry[(size_t)G * 65u + 64u];
    // start_seg's entry state is given, never rewritten
    if (p.conv_pass) {
        bool changed = false;
        sym_sweep_range<true>(p.summary + (size_t)(im.seg_base + lo) * 65u, hi - lo, lane, tabv, pxv,
                              p.entry + (size_t)(im.seg_base + lo) * 65u, lo == im.start_seg ? 1u : 0u, &changed);
        refine_pass_note(p, changed, lane);
        return;
    }
    sym_sweep_range(p.summary + (size_t)(im.seg_base + lo) * 65u, hi - lo, lane, tabv, pxv,
                    p.entry + (size_t)(im.seg_base + lo) * 65u, lo == im.start_seg ? 1u : 0u);
}

// The group levels with FOUR wavefronts per group (calls of a few images, round 6).  A step of these chains is a dependent gather + a
// dozen vector instructions, 0.17 us whatever is done about its loads, and a lone 4K frame's decode spent 64 of them in l1 and 64 in l3
// one after the other (15 + 16 us).  Here a wavefront composes / sweeps a QUARTER of the group: l1 = 16 steps + 4 to compose the
// quarters, l3 = up to 3 steps over the quarters in front + 16 - the same function of the summaries, a third of the depth.
// The per-image level for those calls as PREFIXES (round 6): dec_chain_state_l2m composed its shares, handed a summary from workgroup to
// workgroup (flags, spins), applied 16 shares and swept its share - 59 dependent steps + waits, 19.5 us on a lone 4K frame.  Here a
// wavefront composes its share and STORES the inclusive prefix after every step, the first wavefront does the same over the workgroup's 16
// shares: 26 steps, nothing waits.  The group's entry state is then the image's start state pushed through (workgroups in front)(shares in
// front, one prefix)(groups in front, one prefix) - in dec_chain_state_l3q, every group for itself.
constexpr uint32_t kL2pWgs = 8;
__device__ __forceinline__ void sym_compose_range_prefix(const sym_t* __restrict__ rec, uint32_t count, uint32_t lane, sym_t& P_tab, sym_t& P_px, sym_t* __restrict__ out) {
    if (count == 0u) return;
    SymBatch cur, nxt;
    sym_batch_load(cur, rec, 0u, count, lane);
    for (uint32_t g = 0; g < count; g += kChainBatch) {
        const uint32_t n = min(count - g, (uint32_t)kChainBatch);
        sym_batch_load(nxt, rec, g + kChainBatch, count, lane);
#pragma unroll
        for (int i = 0; i < kChainBatch; ++i) {
            if ((uint32_t)i < n) {
                sym_chain_step(cur.tab[i], batch_px(cur, i), P_tab, P_px);
                out[(size_t)(g + i) * 65u + lane] = P_tab;
                if (lane == 0) out[(size_t)(g + i) * 65u + 64u] = P_px;
            }
        }
        cur = nxt;
    }
}
// the same over words other workgroups of THIS launch have written (through to memory, complete before their arrival was counted): read past the caches
__device__ __forceinline__ void sym_compose_range_prefix_coherent(const sym_t* rec, uint32_t count, uint32_t lane, sym_t& P_tab, sym_t& P_px, sym_t* out) {
    for (uint32_t g = 0; g < count; g += 8u) {
        sym_t t[8], x[8];
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) {
            const uint32_t k = min(g + i, count - 1u);
            t[i] = granule_load(&rec[(size_t)k * 65u + lane]); x[i] = granule_load(&rec[(size_t)k * 65u + 64u]);
        }
#pragma unroll
        for (uint32_t i = 0; i < 8u; ++i) {
            if (g + i < count) {
                sym_chain_step(t[i], uniform_sym(x[i]), P_tab, P_px);
                granule_store(&out[(size_t)(g + i) * 65u + lane], P_tab);
                if (lane == 0) granule_store(&out[(size_t)(g + i) * 65u + 64u], P_px);
            }
        }
    }
}
__global__ __launch_bounds__(64 * kL2Waves) void dec_chain_state_l2p(DecParams p) {
    __shared__ sym_t s_sum[kL2Waves][65];
    if (refine_pass_idle(p)) return;
    if (p.tr_scan && *p.sync_fails != 0u) return;
    const uint32_t img = blockIdx.x / kL2pWgs, k = blockIdx.x % kL2pWgs, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const DecImage im = p.images[img];
    if (im.start_seg >= im.n_active) return;
    const uint32_t gfirst = im.start_seg / kGrp, gend = (im.n_active + kGrp - 1u) / kGrp;
    const uint32_t per = (gend - gfirst + kL2pWgs * kL2Waves - 1u) / (kL2pWgs * kL2Waves);
    const uint32_t lo = min(gfirst + (k * kL2Waves + wave) * per, gend), hi = min(lo + per, gend);
    {
        sym_t P_tab = sym_make(0u, lane, 0u), P_px = sym_make(0u, 64u, 0u);          // identity
        sym_compose_range_prefix(p.grp_summary + (size_t)(im.grp_base + lo) * 65u, hi - lo, lane, P_tab, P_px, p.grp_prefix + (size_t)(im.grp_base + lo) * 65u);
        s_sum[wave][lane] = P_tab;
        if (lane == 0) s_sum[wave][64] = P_px;
    }
    __syncthreads();
    if (wave != 0u) return;
    sym_t P_tab = sym_make(0u, lane, 0u), P_px = sym_make(0u, 64u, 0u);
    sym_compose_range_prefix(&s_sum[0][0], kL2Waves, lane, P_tab, P_px, p.share_prefix + (size_t)((img * kL2pWgs + k) * kL2Waves) * 65u);
    p.l2_sum[(size_t)(img * kL2pWgs + k) * 65u + lane] = P_tab;
    if (lane == 0) p.l2_sum[(size_t)(img * kL2pWgs + k) * 65u + 64u] = P_px;
}
__global__ __launch_bounds__(256) void dec_chain_state_l1q(DecParams p) {
    __shared__ sym_t s_q[4][65];
    if (refine_pass_idle(p)) return;
    if (p.tr_scan && *p.sync_fails != 0u) return;
    const uint32_t G = blockIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 + kGrp <= im.start_seg || j0 >= im.n_active) return;
    const uint32_t lo = max(j0, im.start_seg), hi = min(j0 + kGrp, im.n_active);
    const uint32_t wlo = max(lo, j0 + 16u * wave), whi = min(hi, j0 + 16u * wave + 16u);
    sym_t P_tab = sym_make(0u, lane, 0u), P_px = sym_make(0u, 64u, 0u);          // identity
    sym_compose_range(p.summary + (size_t)(im.seg_base + wlo) * 65u, whi > wlo ? whi - wlo : 0u, lane, P_tab, P_px);
    sym_t* const qs = p.qtr_summary + ((size_t)G * 4u + wave) * 65u;
    qs[lane] = P_tab; s_q[wave][lane] = P_tab;
    if (lane == 0) { qs[64] = P_px; s_q[wave][64] = P_px; }
    __syncthreads();
    if (wave != 0u) return;
    P_tab = sym_make(0u, lane, 0u); P_px = sym_make(0u, 64u, 0u);
    sym_compose_range(&s_q[0][0], 4u, lane, P_tab, P_px);
    if (!p.s3_ctr) {
        p.grp_summary[(size_t)G * 65u + lane] = P_tab;
        if (lane == 0) p.grp_summary[(size_t)G * 65u + 64u] = P_px;
        return;
    }
    // ---- the per-image level on this launch (round 6): what dec_chain_state_l2p does in a launch of its own (10 us on a lone 4K frame, half
    // of it the launch).  The group's summary goes THROUGH to memory and is complete (vmcnt 0) before the group's arrival is counted; the
    // last group of a share to arrive composes the share's inclusive prefixes from the words in memory, counts the share's arrival, and
    // the last share of a workgroup's sixteen does the same one level up.  Nobody waits: whoever arrives last finds everything there.
    granule_store(&p.grp_summary[(size_t)G * 65u + lane], P_tab);
    if (lane == 0) granule_store(&p.grp_summary[(size_t)G * 65u + 64u], P_px);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t gfirst = im.start_seg / kGrp, gend = (im.n_active + kGrp - 1u) / kGrp, g = G - im.grp_base;
    const uint32_t per = (gend - gfirst + kL2pWgs * kL2Waves - 1u) / (kL2pWgs * kL2Waves);
    const uint32_t sh = (g - gfirst) / per, k = sh / kL2Waves;
    const uint32_t slo = gfirst + sh * per, shi = min(slo + per, gend);                     // the share's groups
    uint32_t* const ctr = p.s3_ctr + (size_t)(img * kL2pWgs + k) * (kL2Waves + 1u);            // [0..15] the shares of workgroup k, [16] the workgroup
    uint32_t t = 0u;
    if (lane == 0) t = atomicAdd(&ctr[sh % kL2Waves], 1u);
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)t) + 1u != shi - slo) return;
    P_tab = sym_make(0u, lane, 0u); P_px = sym_make(0u, 64u, 0u);
    sym_compose_range_prefix_coherent(p.grp_summary + (size_t)(im.grp_base + slo) * 65u, shi - slo, lane, P_tab, P_px, p.grp_prefix + (size_t)(im.grp_base + slo) * 65u);
    sym_t* const ss = p.share_sum + (size_t)((img * kL2pWgs + k) * kL2Waves) * 65u;
    granule_store(&ss[(size_t)(sh % kL2Waves) * 65u + lane], P_tab);
    if (lane == 0) granule_store(&ss[(size_t)(sh % kL2Waves) * 65u + 64u], P_px);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // shares of this workgroup that hold groups at all (the image's last workgroup may hold fewer than sixteen)
    const uint32_t wlo_g = gfirst + k * kL2Waves * per;
    const uint32_t nshares = wlo_g >= gend ? 0u : min((uint32_t)kL2Waves, (gend - wlo_g + per - 1u) / per);
    if (lane == 0) t = atomicAdd(&ctr[kL2Waves], 1u);
    if ((uint32_t)__builtin_amdgcn_readfirstlane((int)t) + 1u != nshares) return;
    P_tab = sym_make(0u, lane, 0u); P_px = sym_make(0u, 64u, 0u);
    sym_compose_range_prefix_coherent(ss, nshares, lane, P_tab, P_px, p.share_prefix + (size_t)((img * kL2pWgs + k) * kL2Waves) * 65u);
    granule_store(&p.l2_sum[(size_t)(img * kL2pWgs + k) * 65u + lane], P_tab);
    if (lane == 0) granule_store(&p.l2_sum[(size_t)(img * kL2pWgs + k) * 65u + 64u], P_px);
}
__global__ __launch_bounds__(256) void dec_chain_state_l3q(DecParams p) {
    if (refine_pass_idle(p)) return;
    if (p.tr_scan && *p.sync_fails != 0u) return;
    const uint32_t G = blockIdx.x, lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t img = find_image_by_group(p.images, p.n_images, G);
    const DecImage im = p.images[img];
    const uint32_t j0 = (G - im.grp_base) * kGrp;
    if (j0 + kGrp <= im.start_seg || j0 >= im.n_active) return;
    const uint32_t lo = max(j0, im.start_seg), hi = min(j0 + kGrp, im.n_active);
    const uint32_t wlo = max(lo, j0 + 16u * wave), whi = min(hi, j0 + 16u * wave + 16u);
    if (whi <= wlo) return;
    uint32_t tabv, pxv;
    if (p.grp_prefix) {
        // the group's entry state from the image's start state (given at start_seg) and the prefixes dec_chain_state_l2p left
        const uint32_t gfirst = im.start_seg / kGrp, gend = (im.n_active + kGrp - 1u) / kGrp, g = G - im.grp_base;
        const uint32_t per = (gend - gfirst + kL2pWgs * kL2Waves - 1u) / (kL2pWgs * kL2Waves);
        const uint32_t sh = (g - gfirst) / per, k = sh / kL2Waves, c = sh % kL2Waves, i = (g - gfirst) - sh * per;
        const size_t q0 = (size_t)im.seg_base + im.start_seg;
        tabv = p.entry[q0 * 65u + lane]; pxv = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.entry[q0 * 65u + 64u]);
        sym_t rt[kL2pWgs + 1u], rp[kL2pWgs + 1u];               // all asked for at once: workgroups 0..k-1 (k <= 7), the share prefix, the group prefix
#pragma unroll
        for (uint32_t m = 0; m < kL2pWgs - 1u; ++m) {
            const sym_t* r = p.l2_sum + (size_t)(img * kL2pWgs + min(m, k ? k - 1u : 0u)) * 65u;
            rt[m] = r[lane]; rp[m] = r[64];
        }
        {   const sym_t* r = p.share_prefix + (size_t)((img * kL2pWgs + k) * kL2Waves + (c ? c - 1u : 0u)) * 65u;
            rt[kL2pWgs - 1u] = r[lane]; rp[kL2pWgs - 1u] = r[64]; }
        {   const sym_t* r = p.grp_prefix + (size_t)(G - (i ? 1u : 0u)) * 65u;
            rt[kL2pWgs] = r[lane]; rp[kL2pWgs] = r[64]; }
#pragma unroll
        for (uint32_t m = 0; m < kL2pWgs - 1u; ++m) if (m < k) sym_sweep_step(rt[m], uniform_sym(rp[m]), tabv, pxv);
        if (c != 0u) sym_sweep_step(rt[kL2pWgs - 1u], uniform_sym(rp[kL2pWgs - 1u]), tabv, pxv);
        if (i != 0u) sym_sweep_step(rt[kL2pWgs], uniform_sym(rp[kL2pWgs]), tabv, pxv);
    } else {
        tabv = p.grp_entry[(size_t)G * 65u + lane]; pxv = (uint32_t)__builtin_amdgcn_readfirstlane((int)p.grp_entry[(size_t)G * 65u + 64u]);
    }
    sym_t qt[3], qp[3];                                       // the quarters in front of this wavefront's (all three asked for: the loads do not wait for each other)
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) {
        const sym_t* qs = p.qtr_summary + ((size_t)G * 4u + min(k, wave ? wave - 1u : 0u)) * 65u;
        qt[k] = qs[lane]; qp[k] = qs[64];
    }
#pragma unroll
    for (uint32_t k = 0; k < 3u; ++k) if (k < wave) sym_sweep_step(qt[k], uniform_sym(qp[k]), tabv, pxv);
    // start_seg's entry state is given, never rewritten
    if (p.conv_pass) {
        bool changed = false;
        sym_sweep_range<true>(p.summary + (size_t)(im.seg_base + wlo) * 65u, whi - wlo, lane, tabv, pxv,
                              p.entry + (size_t)(im.seg_base + wlo) * 65u, wlo == im.start_seg ? 1u : 0u, &changed);
        refine_pass_note(p, changed, lane);
        return;
    }
    sym_sweep_range(p.summary + (size_t)(im.seg_base + wlo) * 65u, whi - wlo, lane, tabv, pxv,
                    p.entry + (size_t)(im.seg_base + wlo) * 65u, wlo == im.start_seg ? 1u : 0u);
}

// =====================================================================================
// Chunk records (qoi_decode_core.h "Chunk RECORDS"): dec_transcode writes them once, dec_summarize_rec (P3) and
// dec_segments_rec (P4) read them.
//
// Why: profiles/r01_s8_sq_counters.txt - the two table-bound passes ran at half of the CU's issue rate, each as a
// READER wavefront (byte cursor, chunk table, LUMA expansion: ~45 instructions per chunk step) beside the wavefront that
// owns the colour tables, and the 16 KiB of tables per 64 segments (+ stream ring + record buffers: 39 KiB per pair) held
// a CU to four such pairs.  The reader's work does not depend on pixel state and was done twice (P3 and P4), after P1
// and P2 had walked the same chunks already.  Now the walk that P2 needs anyway (from every segment's true entry
// position) leaves one 32-bit record per chunk; P3 / P4 are single wavefronts of 20 KiB (tables + pixel ring / source
// codes): EIGHT per CU, each stepping through ~35-40 instructions per chunk with its records arriving as 16-byte
// buffer loads a block of eight steps ahead - no LDS ring, no cursor, a wave-uniform loop count.
// =====================================================================================
// PipeReader: the byte source of dec_transcode.  Same LDS ring as LaneReaderT (32 dwords per lane + mirror, [dword][lane]), but
// the refill is a STATIC pipeline three periods deep: every period of four steps every lane issues one 32-byte request (two
// 16-byte loads; lanes whose ring has no room read a dummy line all lanes share) into one of three register sets, and lands
// the set it issued three periods earlier.  The number of loads per period does not depend on the data, so the wait before
// landing is `s_waitcnt vmcnt(N)`: the line asked for twelve steps ago, not the ones asked for since.  With LaneReaderT's
// land-next-period scheme a wavefront waited for HBM almost every period (some lane of 64 always has a fresh request out):
// SQ_WAIT_ANY 55 % of the wavefront cycles (profiles/r02), whatever the ring size or the occupancy.
// Worst case (every chunk five bytes): a period consumes 20 bytes (+ 13 of look-ahead); with 128 bytes of ring, 32-byte
// requests and up to three in flight the unread part never falls below 40 bytes (simulated over all phases).
// DESC: the requests go through ONE raw buffer descriptor per wavefront (base: the stream of its first segment; the lanes' streams
// follow it in memory) with 32-bit offsets: a lane without a request aims outside the descriptor - no memory access, no dummy
// line - and neither the 64-bit address arithmetic nor the clamp to the stream's last granule is needed: two vector instructions
// per period for the two loads instead of twelve (the refill was 9 of the transcoder's 28 vector instructions per chunk).
// Bytes behind a stream's end are never parsed (qoi.h:539), so what a lane reads there - the slack up to the next stream, zeros
// behind the wavefront's last stream - does not matter; the descriptor ends with the 16-byte granule of that last stream's last
// byte, as the plain form's clamp does.  A lane whose stream lies 4 GiB or more behind the base (never with sane strides) cannot
// be addressed: init() says so and the segment goes the way of the segments whose parse did not synchronise (MODE 1, plain form).
template <bool DESC>
struct PipeReaderT {
    static constexpr uint32_t RD = 32, kSlots = RD + 1, kPeriod = 4;
    uint32_t ring;             // LDS byte address of ring[0][lane]
    const uint8_t* abase;      // 32-byte aligned start of the fetched range              (plain form)
    const uint8_t* alast;      // last 16-byte granule that starts before stream + size   (plain form)
    const uint8_t* dummy;      // 16 readable bytes every idle lane loads instead (one line for the whole wavefront; plain form)
    __amdgpu_buffer_rsrc_t rs; // DESC: the wavefront's descriptor
    uint32_t boff;             // DESC: offset of abase in the descriptor
    uint32_t idle;             // DESC: an offset outside the descriptor (wave-uniform)
    uint32_t aoff;             // abase - stream
    uint32_t wr;               // dwords landed, counted from abase
    uint32_t req;              // dwords requested (landed + in flight)
    uint4 set[3][2];
    bool valid[3];
    __device__ __forceinline__ uint4 load16(const uint8_t* p, bool go) const {
        const uint8_t* q = p < alast ? p : alast;                 // a granule past the end is replaced by the last one inside
        return load_global16(go ? q : dummy);
    }
    __device__ __forceinline__ static uint4 as_uint4(u32x4 v) { return make_uint4(v.x, v.y, v.z, v.w); }
    __device__ __forceinline__ void put4(uint32_t at, const uint4& v) {
        const uint32_t a = ring + (at & (RD - 1u)) * 256u;
        lds_u32* q = (lds_u32*)a;
        q[0] = v.x; q[64] = v.y; q[128] = v.z; q[192] = v.w;
        if ((at & (RD - 1u)) == 0u) ((lds_u32*)ring)[RD * 64u] = v.x;         // mirror of slot 0
    }
    // DESC form.  wave_base: the stream of the wavefront's first segment; wave_bytes: from there to the end of the 16-byte granule
    // that holds the last byte of the last stream a lane of the wavefront reads (both wave-uniform); lane_ok: the lane reads at all.
    // Returns false for a lane whose stream the descriptor does not reach.
    // lane_end: one past the 16-byte granule that holds the last byte the lane may read (its stream's end).
    __device__ __forceinline__ bool init_desc(uint32_t ring_addr, const uint8_t* wave_base, unsigned long long wave_bytes, const uint8_t* stream, uint32_t pos0, bool lane_ok,
                                              const uint8_t* lane_end) {
        ring = ring_addr;
        const unsigned long long nrec = wave_bytes < 0xFFFFFFE0ull ? wave_bytes : 0xFFFFFFE0ull;
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)wave_base, 0, (int)(uint32_t)nrec, 0x00020000);
        idle = (uint32_t)nrec;                                             // the first offset outside (and so is idle + 16)
        const uint8_t* p = stream + pos0;
        const uint8_t* a = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)31);
        const unsigned long long d = (unsigned long long)(a - wave_base);
        // the lane's WHOLE read extent must lie inside the descriptor: a stream that begins just below a capped descriptor's end would
        // read zeros behind it (zeros parse as QOI_OP_INDEX chunks, the lane would "synchronise" on them) - it takes MODE 1 instead
        const bool reach = lane_ok && a >= wave_base && lane_end >= a && d + (unsigned long long)(lane_end - a) <= nrec;
        boff = reach ? (uint32_t)d : idle;
        aoff = pos0 - (uint32_t)(p - a);
        u32x4 v[RD / 4u];
#pragma unroll
        for (uint32_t r = 0; r < RD / 4u; ++r) v[r] = __builtin_amdgcn_raw_buffer_load_b128(rs, reach ? boff + 16u * r : idle, 0, 0);
#pragma unroll
        for (uint32_t r = 0; r < RD / 4u; ++r) put4(4u * r, as_uint4(v[r]));
        wr = req = RD;
        valid[0] = valid[1] = valid[2] = false;
        return reach || !lane_ok;
    }
    __device__ __forceinline__ void init(uint32_t ring_addr, const uint8_t* stream, uint32_t pos0, uint32_t size, const uint8_t* dummy16) {
        ring = ring_addr; dummy = dummy16;
        const uint8_t* p = stream + pos0;
        abase = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(p) & ~(uintptr_t)31);
        alast = reinterpret_cast<const uint8_t*>(reinterpret_cast<uintptr_t>(stream + size - 1u) & ~(uintptr_t)15);
        aoff = pos0 - (uint32_t)(p - abase);
        uint4 v[RD / 4u];
#pragma unroll
        for (uint32_t r = 0; r < RD / 4u; ++r) v[r] = load16(abase + 16u * r, true);
#pragma unroll
        for (uint32_t r = 0; r < RD / 4u; ++r) put4(4u * r, v[r]);
        wr = req = RD;
        valid[0] = valid[1] = valid[2] = false;
    }
    __device__ __forceinline__ void peek(uint32_t pos, uint32_t& w32, uint32_t& b5) const {
        const uint32_t rp = pos - aoff;
        const lds_u32* q = (const lds_u32*)(ring + ((rp >> 2) & (RD - 1u)) * 256u);
        const uint32_t d0 = q[0], d1 = q[64];
        const uint32_t sh = (rp & 3u) * 8u;
        w32 = __builtin_amdgcn_alignbit(d1, d0, sh);
        b5 = (d1 >> sh) & 0xFFu;
    }
    // the same, byte 4 left for the caller who needs it: b5 = (hi >> sh) & 0xFF (only a QOI_OP_RGBA's alpha record asks)
    __device__ __forceinline__ void peek4(uint32_t pos, uint32_t& w32, uint32_t& hi, uint32_t& sh) const {
        const uint32_t rp = pos - aoff;
        const lds_u32* q = (const lds_u32*)(ring + ((rp >> 2) & (RD - 1u)) * 256u);
        const uint32_t d0 = q[0]; hi = q[64];
        sh = (rp & 3u) * 8u;
        w32 = __builtin_amdgcn_alignbit(hi, d0, sh);
    }
    // the same for a cursor kept RELATIVE to the ring's origin (rp = pos - aoff): the transcoder's walk keeps it that way and
    // saves the subtraction in every step
    __device__ __forceinline__ void peek4_rel(uint32_t rp, uint32_t& w32, uint32_t& hi, uint32_t& sh) const {
        static_assert(RD == 32u, "five bits of ring slot");
        uint32_t a;                                                       // ring + 256 * ((rp / 4) % 32) in two instructions (hipcc makes it shift, and, add)
        asm("v_bfe_u32 %0, %1, 2, 5\n\tv_lshl_add_u32 %0, %0, 8, %2" : "=&v"(a) : "v"(rp), "v"(ring));
        const lds_u32* q = (const lds_u32*)a;
        const uint32_t d0 = q[0]; hi = q[64];
        sh = rp;                                                         // the caller who needs byte 4 shifts by 8 * (sh & 3)
        w32 = __builtin_amdgcn_alignbyte(hi, d0, rp);                    // v_alignbyte takes the byte count from the low two bits
    }
    template <int S>
    __device__ __forceinline__ void turn_rel(uint32_t rp) { turn<S>(rp + aoff); }
    // one period's memory work for register set S (compile-time 0..2): land what S holds (asked for three periods ago), ask again
    template <int S>
    __device__ __forceinline__ void turn(uint32_t pos) {
        if (valid[S]) {
            // wr is a multiple of eight (init lands RD dwords, every turn eight): the second half sits four slots on without a wrap
            // and is never slot 0
            const uint32_t a = ring + (wr & (RD - 1u)) * 256u;
            lds_u32* q = (lds_u32*)a;
            const uint4 v0 = set[S][0], v1 = set[S][1];
            q[0] = v0.x; q[64] = v0.y; q[128] = v0.z; q[192] = v0.w;
            q[256] = v1.x; q[320] = v1.y; q[384] = v1.z; q[448] = v1.w;
            if ((wr & (RD - 1u)) == 0u) ((lds_u32*)ring)[RD * 64u] = v0.x;        // mirror of slot 0
            wr += 8u;
        }
        const uint32_t space = RD - (req - ((pos - aoff) >> 2));      // dwords neither unread nor on their way
        const bool go = space >= 8u;
        if (DESC) {
            const uint32_t off = go ? boff + req * 4u : idle;
            set[S][0] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0));
            set[S][1] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u, 0, 0));
        } else {
            set[S][0] = load16(abase + (size_t)req * 4u, go);
            set[S][1] = load16(abase + (size_t)req * 4u + 16u, go);
        }
        valid[S] = go;
        req += go ? 8u : 0u;
    }
};

#ifndef QOIMI_TR_WAVES
#define QOIMI_TR_WAVES 4
#endif
#ifndef QOIMI_TR_STORE_AUX
#define QOIMI_TR_STORE_AUX 2                // cache policy of the record stores: nt
#endif
typedef PipeReaderT<false> TransReader;
constexpr uint32_t kTrThreads = 64u * QOIMI_TR_WAVES;
// per tag byte: record template and chunk-table word (QOI_OP_RGBA's length set to 0: visited twice) side by side - ONE 8-byte LDS read per
// step (two 4-byte reads of two arrays: a third of the kernel's LDS cycles were bank conflicts of these reads, profiles/r04_s6_sq_counters_decode.txt);
// entry 256: the null chunk of a lane that is through (no record, no bytes, no pixels)
struct LdsLutT { uint2 e[260]; };

// Parse + P2 + transcode: lane = segment.  Walks every chunk that starts in the segment, writes the chunk records of the
// segment as 16-byte granules (row g of the wavefront's block, null-padded), counts the pixels and leaves the speculative
// slot/alpha transfer (slot_rec).  A QOI_OP_RGBA chunk is visited in two consecutive steps
// (stash record, alpha record); only the second advances the cursor.
//
// MODE 0 - where does the segment's first chunk start?  Chunk length is a function of the first byte (qoi.h:547-575), so
// parse chains that meet stay together.  The lane starts FIVE chains kSyncBytes before its segment, on five consecutive
// bytes: a chunk is at most five bytes long, so the stream's true chain stands on one of them.  Once all five have met -
// in a natural image after a handful of chunks - their common continuation is the true chain whatever the phase was, and
// the first position at or behind the segment start is the entry position: EXACT, not a guess.  The separate parse pass
// (dec_parse_fine + its five-phase records) is then not needed: the lane writes a parse record that holds the same for
// all five phases, and S1 chains those as before (pixel offsets, n_active).  Chains that have not met when they reach
// the segment (runs of equally long multi-byte chunks: QOI_OP_RGBA noise never meets) flag the segment; flagged segments
// get the full five-phase parse (dec_parse_fine, which returns at once when there are none) and are transcoded by MODE 1
// from the entry phase S1 then knows.
// MODE 1 - segments flagged by MODE 0 (all active ones if DecParams::sync_all), entry phase from S1.
constexpr uint32_t kSyncBytes = 64;
constexpr u64 kScanAgg = 1ull << 62, kScanIncl = 2ull << 62;
// (Two ways of leaving the tail walk to dec_transcode<0> were built and measured, both a wash on a lone 4K frame: reading the lane's own
// last rows back in its epilogue - 31.3 + 19.6 us became 38.2 + 13.5; keeping the transfer while walking (dec_transcode<0, TRACK>: four
// vector instructions per record) - 32.4 + 19.8 became 38.5 + 13.8, profiles/r06_s12.  A wavefront that is alone on its SIMD issues one
// instruction every ~8 cycles whatever depends on what: these kernels are bound by the instruction count of their longest wavefront, and
// work moved from one to the other costs there what it saves here.)
constexpr bool kTailsInTranscoder = false;
constexpr uint32_t kScanPxCap = (1u << 29) - 1u;
__device__ __forceinline__ uint32_t scan_sat(uint32_t a, uint32_t b) { const uint32_t s = sat_add(a, b); return s < kScanPxCap ? s : kScanPxCap; }
__device__ __forceinline__ uint32_t slots_then(uint32_t a, uint32_t b) { return slot_pack(slot_compose(slot_unpack(a), slot_unpack(b))); }     // b after a, packed
__device__ __forceinline__ u64 scan_word(const DecParams& p, u64 state, uint32_t px, uint32_t slots) {
    const SlotRec r = slot_unpack(slots);
    const uint32_t s17 = (uint32_t)r.hc | ((uint32_t)r.h_rel << 6) | ((uint32_t)r.h_alpha << 7) | ((uint32_t)r.a_abs << 8) | ((uint32_t)r.ac << 9);
    return state | ((u64)(p.epoch & 0xFFFFu) << 46) | ((u64)s17 << 29) | (u64)(px < kScanPxCap ? px : kScanPxCap);
}
__device__ __forceinline__ uint32_t scan_word_slots(u64 w) {
    const uint32_t s17 = (uint32_t)(w >> 29) & 0x1FFFFu;
    SlotRec r; r.hc = s17 & 63u; r.h_rel = (s17 >> 6) & 1u; r.h_alpha = (s17 >> 7) & 1u; r.a_abs = (s17 >> 8) & 1u; r.ac = (s17 >> 9) & 0xFFu;
    return slot_pack(r);
}
// both inclusive scans of dec_scan_entry in one loop (the two gathers of a round travel together)
__device__ __forceinline__ void wave_scan_px_slots(uint32_t& px, uint32_t& sl, uint32_t lane) {
#pragma unroll
    for (uint32_t d = 1; d < 64u; d <<= 1) {
        const uint32_t up_p = gather_lane(px, lane - d), up_s = gather_lane(sl, lane - d);
        if (lane >= d) { px = sat_add(up_p, px); sl = slot_pack(slot_compose(slot_unpack(up_s), slot_unpack(sl))); }
    }
}

// ---------------------------------------------------------------------------------
// dec_scan_entry's work as the EPILOGUE of dec_transcode<0, .., SPLIT, SCAN> (round 6, calls of a few images at 128-byte segments):
// every lane - half a segment - walks the tail of the records it has just written (its own rows), the workgroup scans pixels and
// slot / alpha transfers over its 256 halves, publishes, looks back over the workgroups in front of it (which took their tickets
// before this one: they are running or done) and the even lanes leave px_off / slot_in / alpha_in of their segments.  One launch and
// two dependent round trips fewer than with dec_scan_entry behind it (24.5 + 20.3 us on a lone 4K frame).
// A call in which some segment did not synchronise is void as a whole: the kernels behind this one look at sync_fails and return.
// ---------------------------------------------------------------------------------
struct ScanShared { uint32_t wpx[4], wsl[4], epx, esl; };
__device__ __forceinline__ void transcode_scan_tail(const DecParams& p, ScanShared& sh, uint32_t gb, uint32_t img, const DecImage& im, uint32_t q, uint32_t half,
                                                    bool part, bool seg_even, uint32_t npix_half, uint32_t mine_half, uint32_t seg_pixels, uint32_t lane, uint32_t wave) {
    const uint32_t per_blk = kTrThreads / 2u;                       // segments per workgroup
    const uint32_t first_blk = im.seg_base / per_blk, blk = gb - first_blk;
    const uint32_t j = q - im.seg_base;
    if (blk == 0u && wave == 0u && im.nseg != 0u) {                // the decoder's start state (qoi.h:533-537) at the image's first segment
        p.entry[(size_t)im.seg_base * 65u + lane] = 0u;
        if (lane == 0) p.entry[(size_t)im.seg_base * 65u + 64u] = kInitPx;
    }
    uint32_t ipx = part ? npix_half : 0u, isl = part ? mine_half : kSlotIdentity;
    wave_scan_px_slots(ipx, isl, lane);
    if (lane == 63u) { sh.wpx[wave] = ipx; sh.wsl[wave] = isl; }
    __syncthreads();
    uint32_t wpx = 0u, wsl = kSlotIdentity, bpx = 0u, bsl = kSlotIdentity;
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t a = sh.wpx[k], b = sh.wsl[k];
        if (k < wave) { wpx = scan_sat(wpx, a); wsl = slots_then(wsl, b); }
        bpx = scan_sat(bpx, a); bsl = slots_then(bsl, b);
    }
    if (wave == 0u) {
        u64* const st = p.scan_status;
        uint32_t epx = 0u, esl = kSlotIdentity;
        if (blk != 0u) {
            if (lane == 0) granule_store(&st[gb], scan_word(p, kScanAgg, bpx, bsl));
            constexpr int kScanWin = 8;
            uint32_t look = blk;
            bool done = false;
            while (!done) {
                u64 w[kScanWin];
#pragma unroll
                for (int k = 0; k < kScanWin; ++k) {
                    const int idx = (int)look - 64 * (k + 1) + (int)lane;
                    w[k] = idx >= 0 ? granule_load(&st[first_blk + (uint32_t)idx]) : scan_word(p, kScanIncl, 0u, kSlotIdentity);
                }
#pragma unroll
                for (int k = 0; k < kScanWin; ++k) {
                    if (done) break;
                    const bool ok = ((uint32_t)(w[k] >> 46) & 0xFFFFu) == (p.epoch & 0xFFFFu) && (w[k] >> 62) != 0ull;
                    const u64 incl = lanes_where(ok && (w[k] >> 62) == 2ull);
                    const int stop = incl ? 63 - (int)__builtin_clzll(incl) : 0;
                    if (lanes_where(!ok && (int)lane >= stop) != 0ull) { __builtin_amdgcn_s_sleep(1); break; }
                    const bool mine_w = (int)lane >= stop;
                    uint32_t vpx = mine_w ? (uint32_t)w[k] & kScanPxCap : 0u, vsl = mine_w ? scan_word_slots(w[k]) : kSlotIdentity;
                    wave_scan_px_slots(vpx, vsl, lane);
                    epx = scan_sat(read_lane(vpx, 63), epx); esl = slots_then(read_lane(vsl, 63), esl);
                    if (incl) done = true;
                    else look -= 64u;
                }
            }
        }
        if (lane == 0) {
            granule_store(&st[gb], scan_word(p, kScanIncl, scan_sat(epx, bpx), slots_then(esl, bsl)));
            sh.epx = epx; sh.esl = esl;
        }
    }
    __syncthreads();
    const uint32_t epx = sh.epx, esl = sh.esl;
    const uint32_t before = scan_sat(scan_sat(epx, wpx), from_lane_below(ipx, 0u));             // pixels in front of this half
    const uint32_t my_off = min(before, im.npx);
    uint32_t slot = hash_px(kInitPx), alpha = kInitPx >> 24;
    slot_apply(slot_unpack(slots_then(slots_then(esl, wsl), from_lane_below(isl, kSlotIdentity))), slot, alpha);
    if (seg_even) { p.px_off[q] = my_off; p.slot_in[q] = (uint8_t)slot; p.alpha_in[q] = (uint8_t)alpha; }
    // n_active: the one segment that starts before the pixel limit while the segment behind it (if any) does not
    const uint32_t next_off = scan_sat(before, seg_pixels);
    if (seg_even && my_off < im.npx && (j + 1u >= im.nseg || next_off >= im.npx)) atomicMax(&p.images[img].n_active, j + 1u);
    if (threadIdx.x == 0 && gb * per_blk + per_blk >= im.seg_base + im.nseg) p.images[img].total_px = min(scan_sat(epx, bpx), im.npx);     // the image's last workgroup
}

// TRACK (round 6 experiment, not instantiated by default - kTailsInTranscoder): the lane also keeps the slot / alpha transfer of its
// segment as it goes (SlotRec: what the chunk that last named a slot absolutely left + the hash shifts behind it), four vector
// instructions per record, so that dec_scan_entry needs no tail walk.  Measured: what the scan saves the transcoder pays (see there).
// SPLIT (round 6, DecParams::tr_split): two lanes per segment, each walks half of it from a run-up of its own; the even lane writes
// the segment's outputs (the odd lane's values come over by a lane shift).
template <int MODE, bool TRACK = false, bool SPLIT = false, bool SCAN = false>
__global__ __launch_bounds__(kTrThreads) void dec_transcode(DecParams p) {
    __shared__ uint32_t s_ring[QOIMI_TR_WAVES][TransReader::kSlots * 64];
    __shared__ LdsLutT s_lut;
    __shared__ ScanShared s_scan;
    __shared__ uint32_t s_gb;
    static_assert(!SCAN || (SPLIT && MODE == 0), "the scan rides on the two-lane form of the first transcoder pass");
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = lane_id();
    if (MODE == 1 && !p.sync_all && *p.sync_fails == 0u) return;
    if (MODE == 1 && !p.sync_all) {                        // (a workgroup without a flagged segment returns before it builds its table: see dec_parse_fine)
        const uint32_t t = blockIdx.x * kTrThreads + threadIdx.x;
        if (!__syncthreads_or((t < p.total_segs && p.sync_fail[t] != 0u) ? 1 : 0)) return;
    }
    if (SCAN) {
        // the workgroup's place in the call by ticket (start order: what its look-back waits for has started); the answer travels while the table is built
        if (threadIdx.x == 0) s_gb = atomicAdd(p.scan_ticket, 1u);
        if (blockIdx.x == 0u && threadIdx.x < p.n_images) p.first_bad[threadIdx.x] = 0xFFFFFFFFu;
    }
    for (uint32_t b = threadIdx.x; b < 256u; b += kTrThreads) {
        uint32_t d, i; lut_entry(b, d, i);
        // the transcoder's own info word: bits 0..2 chunk length (0 for QOI_OP_RGBA: visited twice), bits 8..15 all ones for
        // QOI_OP_LUMA (ANDed onto the chunk bytes it leaves the second byte there, nothing for every other chunk), bits 30..31 class
        // ... bits 16..21 the pixels of the chunk (0 for QOI_OP_RGBA, whose records count for themselves)
        const uint32_t tplb = rec_template(b);
        s_lut.e[b] = make_uint2(tplb, (b == 0xFFu ? 0u : (i & 7u)) | ((i >> 28) & 1u ? 0x0000FF00u : 0u) | (i & 0xC0000000u) | (b == 0xFFu ? 0u : rec_pixels(tplb) << 16));
    }
    if (threadIdx.x < 4u) s_lut.e[256u + threadIdx.x] = make_uint2(0u, 0u);
    __syncthreads();
    const uint32_t gb = SCAN ? s_gb : blockIdx.x;          // (s_gb: written before the barrier above)
    const uint32_t T = gb * kTrThreads + threadIdx.x;
    const uint32_t q = SPLIT ? T >> 1 : T, half = SPLIT ? T & 1u : 0u;
    bool have = q < p.total_segs;
    uint32_t img;
    DecImage im;
    const uint32_t q_img = SCAN ? gb * (kTrThreads / 2u) : (have ? q : 0u);      // SCAN: one image per workgroup (every image begins on a multiple of 256 segments), the same for every lane
    if (MODE == 0 && p.tab_in_args) {
        // the table travels in the kernel arguments (calls of a few images): this launch reads it from there, its first lanes leave the
        // device copy for the kernels that follow
        const uint32_t qq = q_img;
        img = (p.n_images > 1u && qq >= p.tab4[1].seg_base ? 1u : 0u) + (p.n_images > 2u && qq >= p.tab4[2].seg_base ? 1u : 0u) + (p.n_images > 3u && qq >= p.tab4[3].seg_base ? 1u : 0u);
        im = img == 0u ? p.tab4[0] : img == 1u ? p.tab4[1] : img == 2u ? p.tab4[2] : p.tab4[3];
        if (blockIdx.x == 0u && threadIdx.x < p.n_images)
            p.images[threadIdx.x] = threadIdx.x == 0u ? p.tab4[0] : threadIdx.x == 1u ? p.tab4[1] : threadIdx.x == 2u ? p.tab4[2] : p.tab4[3];
    } else {
        img = find_image(p.images, p.n_images, q_img);
        im = p.images[img];
    }
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    if (MODE == 1) have = have && j >= im.start_seg && j < im.n_active && (p.sync_all || p.sync_fail[q] != 0u);
    if (MODE == 0 && have && j >= im.nseg) {               // a padding segment between two images (DecParams::fused): nothing to read, nothing flagged
        if (half == 0u) { p.sync_fail[q] = 0; p.rec_gran[q] = 0u; }
        have = false;
    }
    const uint32_t sbase = (uint32_t)kHeaderBytes + j * p.seg_bytes;        // the segment
    const uint32_t hbytes = SPLIT ? p.seg_bytes >> 1 : p.seg_bytes;
    const uint32_t base = sbase + half * hbytes;                            // this lane's part of it
    if (SPLIT && have && base >= im.chunks_end) have = false;              // the stream ends in the segment's first half (the even lane always has bytes: j < nseg)
    if (!lanes_where(have)) {
        if (SCAN) transcode_scan_tail(p, s_scan, gb, img, im, q, half, false, false, 0u, kSlotIdentity, 0u, lane, wave);      // (the workgroup's barriers are met by every wavefront)
        return;
    }
    const uint32_t end = min(base + hbytes, im.chunks_end);
    const uint32_t lut_base = lds_addr_of(&s_lut.e[0]);
    PipeReaderT<MODE == 0> R;                 // MODE 0: requests through a descriptor; MODE 1 (rare): plain pointers, reaches anything
    uint32_t pos;
    bool failed = false;
    const uint8_t* dummy16 = reinterpret_cast<const uint8_t*>(p.sync_fail);     // any 16 readable bytes: what lanes without a request load
    if (MODE == 1) {
        pos = base + (have ? p.entry_phase[q] : 0u);
        R.init(lds_addr_of(&s_ring[wave][lane]), p.streams + im.stream_off, pos, im.chunks_end + kTrailerBytes, dummy16);
    } else {
        // ---- look-back synchronisation ------------------------------------------------------------------------------
        // (segments of up to 256 bytes - calls of a few images - start their chains 16 bytes back (32 until the second run-up existed): the chains of natural content meet within a
        // dozen bytes, those of noise never do, and at 128-byte segments the 64-byte run-up was a third of the lane's walk; a lane whose
        // chains have not met takes the five-phase parse as ever)
#ifndef QOIMI_SYNC_SHORT
#define QOIMI_SYNC_SHORT 16u             // (round 6, with the second run-up behind it: 16 / 24 / 32 bytes = 26.3 / 26.9 / 28.0 us for a 4K photograph's transcoder, profiles/r06_s43_short_len.txt)
#endif
        const uint32_t back = p.seg_bytes <= 256u ? QOIMI_SYNC_SHORT : kSyncBytes;
        const bool from_start = base <= (uint32_t)kHeaderBytes + back;            // the stream's first chunk is in reach: one chain from byte 14
        const uint32_t t0 = from_start ? (uint32_t)kHeaderBytes : base - back;
        // the wavefront's descriptor: from the stream of its first segment to the last granule of the stream of its last one
        const uint8_t* const my_stream = p.streams + im.stream_off;
        const u64 hv = lanes_where(have);
        const int last = 63 - __builtin_clzll(hv);                                                  // (hv != 0 here)
        const uintptr_t my_end = ((reinterpret_cast<uintptr_t>(my_stream) + im.chunks_end + kTrailerBytes - 1u) & ~(uintptr_t)15) + 16u;
        const uintptr_t last_end = ((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_end, last)) |
                                   ((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_end >> 32), last) << 32);
        const int first = __builtin_ctzll(hv);
        const uintptr_t first_stream = ((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)reinterpret_cast<uintptr_t>(my_stream), first)) |
                                       ((uintptr_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(reinterpret_cast<uintptr_t>(my_stream) >> 32), first) << 32);
        // (base: the first stream aligned DOWN to 32 bytes - the lanes' 32-byte aligned requests then never start in front of it, whatever
        // the alignment of the caller's streams)
        const uintptr_t wave_base = first_stream & ~(uintptr_t)31;
        const bool reach = R.init_desc(lds_addr_of(&s_ring[wave][lane]), reinterpret_cast<const uint8_t*>(wave_base), (unsigned long long)(last_end - wave_base),
                                       my_stream, t0, have, reinterpret_cast<const uint8_t*>(my_end));
        ParseState s; parse_init(s, t0);
        if (from_start) { s.p1 = s.p2 = s.p3 = s.p4 = t0; }
        uint32_t m = t0;
        bool merged = from_start;
        bool going = have && reach && m < base;
        auto sync_steps = [&]() {
#pragma unroll
          for (uint32_t u = 0; u < TransReader::kPeriod; ++u) {
            if (going) {
                uint32_t w32, b5; R.peek(m, w32, b5);
                const uint32_t b1 = w32 & 0xFFu;
                if (!merged) {
                    parse_step(s, m, b1);
                    m = parse_front(s);
                    merged = s.p0 == s.p1 && s.p1 == s.p2 && s.p2 == s.p3 && s.p3 == s.p4;
                } else {
                    m += len_of(b1);
                }
                going = m < base;
            }
          }
        };
        while (lanes_where(going)) {                              // three periods per turn of the register sets (PipeReader)
            R.template turn<0>(m); sync_steps();
            R.template turn<1>(m); sync_steps();
            R.template turn<2>(m); sync_steps();
        }
        bool reach2 = true;
        if (kSyncRetryBytes != 0u && lanes_where(have && reach && !merged) != 0ull) {
            // Chains that have not met within the run-up (a Kodak-like 4K photograph: 7 of 67 000 segments at 32 bytes) get ONE more, longer one
            // before the segment is handed to the five-phase parse - which costs a call of a few images its single-pass path (a wait, the parse,
            // everything again through the chains) and any call a second walk over the segment under plain loads.  Only wavefronts that hold such
            // a lane pay: the reader starts anew, kSyncRetryBytes back for those lanes, where it stands for the others.
            const bool again = have && reach && !merged;
            const bool from_start2 = base <= (uint32_t)kHeaderBytes + kSyncRetryBytes;
            const uint32_t t1 = again ? (from_start2 ? (uint32_t)kHeaderBytes : base - kSyncRetryBytes) : m;
            reach2 = R.init_desc(lds_addr_of(&s_ring[wave][lane]), reinterpret_cast<const uint8_t*>(wave_base), (unsigned long long)(last_end - wave_base),
                                 my_stream, t1, have, reinterpret_cast<const uint8_t*>(my_end));
            if (again) {
                parse_init(s, t1);
                if (from_start2) { s.p1 = s.p2 = s.p3 = s.p4 = t1; }
                m = t1; merged = from_start2;
            }
            going = again && reach2 && m < base;
            while (lanes_where(going)) {
                R.template turn<0>(m); sync_steps();
                R.template turn<1>(m); sync_steps();
                R.template turn<2>(m); sync_steps();
            }
        }
        failed = have && (!merged || !reach || !reach2);
        if (SPLIT) {                                          // either half: the whole segment takes the five-phase parse
            const int other = __shfl_xor((int)failed, 1);     // (every lane asks: a lane that skipped the exchange would hand its neighbour nothing)
            failed = failed || other != 0;
        }
        pos = m;                                              // merged: the first chunk start at or behind the segment start
        // the ring was refilled for the front position; the walk below continues the same byte stream
        const u64 fails = lanes_where(failed && half == 0u);
        if (have && half == 0u) p.sync_fail[q] = failed ? 1 : 0;
        if (fails != 0 && lane == (uint32_t)__builtin_ctzll(fails)) atomicAdd(p.sync_fails, (uint32_t)__builtin_popcountll(fails));
    }
    bool active = have && !failed && pos < end;
    // The walk keeps its cursor relative to the ring's origin (rp) and forms the table offset of the byte under it with one
    // SDWA shift: 20 vector instructions per chunk in the common path (23 with an absolute cursor, round 2).
    uint32_t rp = pos - R.aoff;
    const uint32_t end_rel = end - R.aoff;
    uint32_t three = 3u;
    asm volatile("" : "+v"(three));                                      // the shift count of byte0_times8 lives in a VGPR (SDWA takes no literal)
    auto byte0_times8 = [&](uint32_t w) { uint32_t r; asm("v_lshlrev_b32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(three), "v"(w)); return r; };
    typedef uint32_t u32x2v __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) const u32x2v lds_u32x2;
    uint32_t w32, b5hi, b5sh; R.peek4_rel(rp, w32, b5hi, b5sh);
    uint32_t tpl, info;                        // of the chunk under the cursor; the null entry once the lane is through
    {   const uint32_t i8 = byte0_times8(w32);
        const u32x2v te = *(lds_u32x2*)(lut_base + (active ? i8 : 2048u)); tpl = te.x; info = te.y; }
    uint32_t a_abs = 0u, a_last = 0u;            // a QOI_OP_RGBA occurred in the segment / the alpha of the last one (for dec_slot_tails)
    uint32_t f_hc = 0u, f_rel = 1u, f_alpha = 0u, f_stash = 0u;      // TRACK: the transfer so far (slot_init: identity)
    uint32_t pend = 0u;                          // 1 / 2: the first record of the QOI_OP_RGBA / QOI_OP_RGB chunk under the cursor is out
    bool any_pend = false;
    uint32_t npix = 0u;
    // granule row g of this wavefront's 64 segments: one contiguous KiB.  Stored through a descriptor over the block's rows with a
    // 32-bit offset that moves on by a row per granule (the 64-bit address of every store was four vector instructions)
    const uint32_t rec_blk = SPLIT ? (gb * kTrThreads + wave * 64u) >> 7 : gb * QOIMI_TR_WAVES + wave;      // the block of 64 segments the wavefront's segments lie in
    const __amdgpu_buffer_rsrc_t rs_rec = __builtin_amdgcn_make_buffer_rsrc((void*)(p.recs + (size_t)rec_blk * p.rec_rows * 256u), 0,
                                                                             (int)(p.rec_rows * 1024u), 0x00020000);
    const uint32_t roff0 = (SPLIT ? (q & 63u) : lane) * 16u + (SPLIT ? half * p.tr_rows_half * 1024u : 0u);
    uint32_t roff = roff0;                                               // byte offset of the lane's next granule
    auto granule_steps = [&]() {
        {
            const bool live = active;                                    // the granule holds at least one record of this lane
            uint32_t rr[4];
#pragma unroll
            for (uint32_t u = 0; u < 4u; ++u) {
                const uint32_t c_info = info;
                uint32_t rec = tpl;                                      // null record once the lane is through (LUT entry 256)
                uint32_t adv = lut_len(c_info);
                // byte-wise delta of a relative chunk: table part + the second byte of a LUMA chunk (qoi.h:566-571); the info word
                // keeps that byte (bits 8..15 all ones) for QOI_OP_LUMA only
                const uint32_t wm = w32 & c_info;
                const uint32_t er = __builtin_amdgcn_ubfe(wm, 12, 4), eb = __builtin_amdgcn_ubfe(wm, 8, 4);
                add_byte0(rec, er); add_byte2_from0(rec, eb);
                if (lanes_where(lut_hi(c_info)) != 0 || any_pend) {     // QOI_OP_RGB / QOI_OP_RGBA somewhere in the wavefront (rare in natural images)
                    uint32_t cnt = (rec >> 24) & 63u;                    // pixels of the chunk (qoi.h:573-575); the stash marker is set right below
                    // These chunks leave a PAIR of records that begins on an even record index of the segment (u is a compile-time
                    // constant here): QOI_OP_RGBA = (stash half, alpha half), QOI_OP_RGB = (the record, a null record); a chunk met on
                    // an odd index leaves a null record first.  The chunk stays under the cursor until its second record is out.  So
                    // P3 / P4 can take RGBA-dense content - noise, sprites with many alpha levels - a pair at a time, whatever mix of
                    // the two ops the lanes hold (dec_summarize_rec, dec_segments_rec: "a block of pairs"); a null record is a no-op
                    // anywhere but in front of a stream's first chunk, and index 0 is even.
                    const bool hi = lut_hi(c_info), lo = lut_lo(c_info);
                    const bool rgba = hi && lo;
                    const uint32_t rgb = (w32 >> 8) & 0x00FFFFFFu;
                    const uint32_t b5 = (b5hi >> ((b5sh & 3u) * 8u)) & 0xFFu;                // chunk byte 4: the alpha of a QOI_OP_RGBA
                    if ((u & 1u) == 0u) {                                                    // a pair begins (no lane has one pending here)
                        rec = hi ? (rec | rgb) : rec;                                        // rec still is the class-2 template: stash marker / one pixel
                        adv = hi ? 0u : adv;
                        cnt = rgba ? 0u : cnt;
                        pend = hi ? (lo ? 1u : 2u) : 0u;
                    } else {
                        const bool second_rgba = pend == 1u, second_rgb = pend == 2u, stall = hi && pend == 0u;
                        rec = second_rgba ? rec_make(3u, 1u, b5) : ((second_rgb || stall) ? 0u : rec);
                        adv = second_rgba ? 5u : (second_rgb ? 4u : (stall ? 0u : adv));
                        cnt = second_rgba ? 1u : ((second_rgb || stall) ? 0u : cnt);
                        a_abs = second_rgba ? 1u : a_abs;
                        a_last = second_rgba ? b5 : a_last;
                        pend = 0u;
                    }
                    any_pend = lanes_where(pend != 0u) != 0;
                    npix += cnt;
                    if (TRACK) {                                         // the record as it goes out (tail_step's cases, forwards)
                        const uint32_t cls = rec >> 30, pay = rec & 0x00FFFFFFu;
                        const bool is_stash = cls == 2u && ((rec >> 24) & 63u) == kRecStash;
                        const uint32_t lh = __builtin_amdgcn_udot4(pay, 0x00070503u, 0u, false);
                        if (cls == 0u) f_hc += lh;
                        else if (cls == 1u) { f_hc = rec & 63u; f_rel = 0u; f_alpha = 0u; }
                        else if (is_stash) f_stash = pay;
                        else if (cls == 2u) { f_hc = lh + (a_abs ? 11u * a_last : 0u); f_rel = 0u; f_alpha = a_abs ? 0u : 1u; }
                        else { f_hc = __builtin_amdgcn_udot4(f_stash, 0x00070503u, 0u, false) + 11u * (rec & 0xFFu); f_rel = 0u; f_alpha = 0u; }
                    }
                } else {
                    add_dword_byte2(npix, c_info);                       // the chunk's pixels straight from the table word
                    if (TRACK) {                                         // relative (RUN and the null record: nothing added) or INDEX
                        const bool idx = rec >= 0x40000000u;
                        f_hc = idx ? rec : f_hc + __builtin_amdgcn_udot4(rec & 0x00FFFFFFu, 0x00070503u, 0u, false);
                        f_rel = idx ? 0u : f_rel; f_alpha = idx ? 0u : f_alpha;
                    }
                }
                const uint32_t nrp = rp + adv;
                uint32_t nw32; R.peek4_rel(nrp, nw32, b5hi, b5sh);
                active = active && nrp < end_rel;
                const uint32_t i8 = byte0_times8(nw32);                  // (outside the select: a call in a ?: arm is a branch)
                const u32x2v te = *(lds_u32x2*)(lut_base + (active ? i8 : 2048u));
                const uint32_t ntpl = te.x, ninfo = te.y;
                rr[u] = rec;
                rp = nrp; w32 = nw32; tpl = ntpl; info = ninfo;
            }
            // non-temporal: 13.7 GB of records per 412 frames must not sweep the stream lines out of the L2 between a lane's four
            // 32-byte requests to one 128-byte line (with plain stores FETCH_SIZE was 3.8 x the stream bytes, now 2.5 x)
            if (live) { u32x4 v; v.x = rr[0]; v.y = rr[1]; v.z = rr[2]; v.w = rr[3]; __builtin_amdgcn_raw_buffer_store_b128(v, rs_rec, roff, 0, QOIMI_TR_STORE_AUX); roff += 1024u; }
        }
    };
    while (lanes_where(active)) {                                 // a period = one granule of four steps; three periods per turn of the register sets
        R.template turn_rel<0>(rp); granule_steps();
        R.template turn_rel<1>(rp); granule_steps();
        R.template turn_rel<2>(rp); granule_steps();
    }
    pos = rp + R.aoff;
    uint32_t gran_word = (roff - roff0) >> 10;                          // granules written
    const bool part = have && !failed;                                  // this lane walked its half
    const uint32_t npix_half = npix;
    uint32_t mine_half = kSlotIdentity;
    if (SCAN) {
        // the half's slot / alpha transfer from the tail of its own rows (tail_step; the rows are this lane's stores, acknowledged: vmcnt 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t n_rows = part ? gran_word : 0u;
        TailState t; tail_init(t);
        t.found = n_rows != 0u ? 0u : 1u;
        uint32_t most = n_rows;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const uint32_t v = (uint32_t)__shfl_xor((int)most, o); most = v > most ? v : most; }
        for (uint32_t i = 0; i < most && lanes_where(t.found == 0u) != 0; i += 4u) {
            u32x4 v[4];
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k) v[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_rec, i + k < n_rows ? roff0 + 1024u * (n_rows - 1u - i - k) : 0x7FFFFFF0u, 0, 0);
#pragma unroll
            for (uint32_t k = 0; k < 4u; ++k)
                if (i + k < n_rows) { tail_step(t, v[k].w, a_abs, a_last); tail_step(t, v[k].z, a_abs, a_last); tail_step(t, v[k].y, a_abs, a_last); tail_step(t, v[k].x, a_abs, a_last); }
        }
        if (n_rows == 0u) t.found = 0u;
        if (part) mine_half = slot_pack(tail_finish(t, a_abs, a_last));
    }
    if (SPLIT) {
        // the odd lane's half over to the even lane, which writes the segment's outputs (a lane that took no part holds nothing: no
        // granules, no pixels, no QOI_OP_RGBA, and `pos` is not looked at)
        const uint32_t n1 = (uint32_t)__shfl_down((int)(have ? gran_word : 0u), 1), px1 = (uint32_t)__shfl_down((int)(have ? npix : 0u), 1);
        const uint32_t ab1 = (uint32_t)__shfl_down((int)(have ? a_abs : 0u), 1), al1 = (uint32_t)__shfl_down((int)a_last, 1);
        const uint32_t pos1 = (uint32_t)__shfl_down((int)pos, 1), hv1 = (uint32_t)__shfl_down((int)(have ? 1u : 0u), 1);
        if (half == 0u) {
            gran_word |= n1 << 16; npix += px1;
            a_last = ab1 ? al1 : a_last; a_abs |= ab1;
            pos = hv1 ? pos1 : pos;
        }
        have = have && half == 0u;
    }
    if (have && !failed) {
        p.rec_gran[q] = gran_word;
        SlotRec r; r.hc = 0; r.h_rel = 0; r.h_alpha = 0; r.a_abs = (uint8_t)a_abs; r.ac = (uint8_t)a_last;      // dec_slot_tails / dec_scan_entry completes it
        if (TRACK) { r.hc = (uint8_t)(f_hc & 63u); r.h_rel = (uint8_t)f_rel; r.h_alpha = (uint8_t)f_alpha; }
        p.slot_rec[q] = r;
        if (MODE == 0) {
            // parse record for S1: the same whatever entry phase S1 asks for - it only ever asks for the true one
            const uint32_t nominal = sbase + p.seg_bytes;
            const uint32_t e = pos > nominal ? pos - nominal : 0u;      // bytes the last chunk reaches into the next segment
            ParseRec pr;
            pr.exit_phase = e * (1u | (1u << 3) | (1u << 6) | (1u << 9) | (1u << 12));
#pragma unroll
            for (int k = 0; k < 5; ++k) pr.pixels[k] = npix;
            p.parse[q] = pr;
        }
    } else if (have && MODE == 0) {
        p.rec_gran[q] = 0u;
    }
    if (SCAN) transcode_scan_tail(p, s_scan, gb, img, im, q, half, part, have, npix_half, mine_half, npix, lane, wave);      // (have: now the even lane of a segment that exists)
}

// Record source of a P3 / P4 wavefront: lane l reads the granules of segment 64 * block + l through a raw buffer descriptor
// over the granule rows of the block (row g: 64 lanes x 16 bytes, contiguous).  A lane that is through (or has no segment) asks for an offset outside the
// descriptor and gets zeros - the null record - so the loops never mask lanes off.
// cache policy bits of the record loads / P4's pixel bursts (buffer instructions: 1 = sc0, 2 = nt, 16 = sc1).
// The bursts are WRITE-THROUGH (sc0 sc1): with plain stores the L2 sent 1.26 x the pixels' bytes to the fabric (PMC WRITE_SIZE,
// 1024 x 4K: 42.0 GB for 34.0) - a 128-byte line gets its two 64-byte bursts ~20 steps apart and partly dirty lines go out more
// than once; written through, every burst leaves once: 1.002 x, dec_segments_rec 4.80 -> 4.64 ms per 256 frames (nt: 1.03 x but
// 4.90 ms).
#ifndef QOIMI_REC_LOAD_AUX
#define QOIMI_REC_LOAD_AUX 0
#endif
#ifndef QOIMI_P4_STORE_AUX
#define QOIMI_P4_STORE_AUX 17
#endif
struct RecSource {
    __amdgpu_buffer_rsrc_t rs;
    uint32_t off;          // byte offset of the lane's column in a granule row
    uint32_t n_gran;       // granules of the lane
    uint32_t n0, skip;     // DecParams::tr_split: granules [0, n0) lie in rows [0, n0), the rest `skip` bytes further on (rows from tr_rows_half)
    static constexpr uint32_t kNowhere = 0x7FFFFFF0u;
    // grans: the segment's rec_gran word
    __device__ __forceinline__ void init(const DecParams& p, uint32_t block64, uint32_t lane, uint32_t grans) {
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(p.recs + (size_t)block64 * p.rec_rows * 256u), 0, (int)(p.rec_rows * 1024u), 0x00020000);
        off = lane * 16u;
        if (p.tr_split) { n0 = grans & 0xFFFFu; n_gran = n0 + (grans >> 16); skip = (p.tr_rows_half - n0) * 1024u; }
        else { n0 = grans; n_gran = grans; skip = 0u; }
    }
    __device__ __forceinline__ uint32_t row_off(uint32_t g) const { return g < n_gran ? off + 1024u * g + (g < n0 ? 0u : skip) : kNowhere; }
    __device__ __forceinline__ u32x4 granule(uint32_t g) const {
        return __builtin_amdgcn_raw_buffer_load_b128(rs, row_off(g), 0, QOIMI_REC_LOAD_AUX);
    }
    // the same as a streaming (non-temporal) load: dec_summarize_rec, -3 %; dec_segments_rec is 3 % slower with it
    __device__ __forceinline__ u32x4 granule_nt(uint32_t g) const {
        return __builtin_amdgcn_raw_buffer_load_b128(rs, row_off(g), 0, 2);
    }
};
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const uint32_t t = (uint32_t)__shfl_xor((int)v, o); v = t > v ? t : v; }
    return __builtin_amdgcn_readfirstlane(v);
}


// P2 on records: the slot/alpha transfer of every segment from the TAIL of its records (tail_step, qoi_decode_core.h).
// One wavefront per 64 segments reads their granule rows from the last one backwards until every lane has met a chunk that
// names a slot absolutely - two or three rows in natural images.
__global__ __launch_bounds__(64) void dec_slot_tails(DecParams p) {
    const uint32_t lane = lane_id();
    const uint32_t q = blockIdx.x * 64u + lane;
    bool have = q < p.total_segs;
    const uint32_t img = find_image(p.images, p.n_images, have ? q : 0u);
    const DecImage im = p.images[img];
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    have = have && j >= im.start_seg && j < im.n_active;
    if (!lanes_where(have)) return;
    const SlotRec in = have ? p.slot_rec[q] : SlotRec{0, 0, 0, 0, 0};
    RecSource S; S.init(p, blockIdx.x, lane, have ? p.rec_gran[q] : 0u);
    TailState t; tail_init(t);
    t.found = (have && S.n_gran != 0u) ? 0u : 1u;                         // nothing to scan: the identity transfer (h_rel = 1, hc = 0)
    const bool empty = have && S.n_gran == 0u;
    // Every lane walks ITS OWN rows from its last one down, two rows per turn (both loads in flight together), while any lane has
    // not met its anchor.  (Round 3 walked a common row index down from the wavefront's LARGEST row count: the lanes' counts differ
    // by tens of rows, every one of them a dependent load - 0.63 ms per 1024 frames for three useful rows per lane.)
    const uint32_t most = wave_max_u32(S.n_gran);
    for (uint32_t i = 0; i < most && lanes_where(t.found == 0u) != 0; i += 2u) {
        const uint32_t g0 = S.n_gran - 1u - i, g1 = S.n_gran - 2u - i;    // (wrap past row 0: far beyond n_gran, granule() returns zeros)
        const u32x4 v0 = S.granule(g0), v1 = S.granule(g1);
        if (i < S.n_gran) { tail_step(t, v0.w, in.a_abs, in.ac); tail_step(t, v0.z, in.a_abs, in.ac); tail_step(t, v0.y, in.a_abs, in.ac); tail_step(t, v0.x, in.a_abs, in.ac); }
        if (i + 1u < S.n_gran) { tail_step(t, v1.w, in.a_abs, in.ac); tail_step(t, v1.z, in.a_abs, in.ac); tail_step(t, v1.y, in.a_abs, in.ac); tail_step(t, v1.x, in.a_abs, in.ac); }
    }
    if (empty) t.found = 0u;
    if (have) p.slot_rec[q] = tail_finish(t, in.a_abs, in.ac);
}

// ---------------------------------------------------------------------------------
// dec_scan_entry (round 6): S1, the tails of P2 and S2 in ONE single-pass kernel, for calls of a few images.
//
// A lone 4K frame at 128-byte segments is 80 000 segments, and what stood between dec_transcode<0> and P3 was ten launches -
// dec_parse_fine (nothing to do), S1 x 3, dec_transcode<1> (nothing to do), dec_slot_tails, S2 x 3 - of which six did a microsecond
// of work under a launch floor of 4.7 us each: 63 of the call's 228 us (profiles/r06_s1_single_dec_timeline.txt).  Both chains carry a
// few bits per segment - pixels before the segment (a sum), slot / alpha transfer (SlotRec, closed under composition) - so they are
// a decoupled look-back scan (the encoder's placement, qoi_encode.hip): a workgroup of 256 segments scans its own values, publishes
// the aggregate as ONE tagged 8-byte word, looks back over the words of the workgroups in front of it 64 at a time until it meets
// an inclusive one, publishes its own inclusive word and writes px_off / slot_in / alpha_in of its segments.  Workgroups take their
// place by ticket (start order): whatever a workgroup waits for has started.  The word's tag is the call's number: nothing is zeroed
// per call (qoimi_decode_batch zeroes the words when their arena is allocated and when the 16-bit tag wraps).
// Only valid when dec_transcode<0> synchronised EVERY segment (sync_fails == 0: encoder-made streams of natural content); otherwise the
// kernel returns at once, the host sees the counter behind the round and continues with the five-phase parse and the three-level chains.
//   word: bits 0..28 pixels (saturating at 2^29 - 1 > QOI_PIXELS_MAX), 29..45 slot transfer, 46..61 tag, 62..63 state
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kScanSegs) void dec_scan_entry(DecParams p) {
    __shared__ uint32_t s_wpx[4], s_wsl[4], s_epx, s_esl, s_blk;
    const uint32_t lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (blockIdx.x == 0u && threadIdx.x < p.n_images) p.first_bad[threadIdx.x] = 0xFFFFFFFFu;     // (whatever follows: dec_fill looks at it)
    if (*p.sync_fails != 0u) return;                       // (every workgroup: the counter stands since dec_transcode<0> ended)
#ifndef QOIMI_SCAN_EXP
#define QOIMI_SCAN_EXP 0              // timing-only builds (tools/measure): 1 no tail walk, 2 no look-back, 4 no ticket - wrong pixels
#endif
    if ((QOIMI_SCAN_EXP & 4) == 0) { if (threadIdx.x == 0) s_blk = atomicAdd(p.scan_ticket, 1u); } else s_blk = blockIdx.x;
    __syncthreads();
    const uint32_t gb = s_blk;                             // this workgroup's place: segments 256 gb ... of the call
    const uint32_t q0 = gb * kScanSegs, q = q0 + threadIdx.x;
    const uint32_t img = find_image(p.images, p.n_images, q0);       // every image begins on a multiple of 256 segments: one image per workgroup
    const DecImage im = p.images[img];
    const uint32_t j = q - im.seg_base;
    const bool have = j < im.nseg;
    const uint32_t first_blk = im.seg_base / kScanSegs, blk = gb - first_blk;
    if (blk == 0u && wave == 0u && im.nseg != 0u) {        // the decoder's start state (qoi.h:533-537) at the image's first segment
        p.entry[(size_t)im.seg_base * 65u + lane] = 0u;
        if (lane == 0) p.entry[(size_t)im.seg_base * 65u + 64u] = kInitPx;
    }
    // ---- the segment's own values: its pixels (dec_transcode<0> counted them) and its slot / alpha transfer from the TAIL of its records (dec_slot_tails)
    const uint32_t npix = have ? p.parse[q].pixels[0] : 0u;
    const SlotRec in = have ? p.slot_rec[q] : SlotRec{0, 0, 0, 0, 0};
    RecSource S; S.init(p, (q0 >> 6) + wave, lane, have ? p.rec_gran[q] : 0u);
    TailState t; tail_init(t);
    t.found = (have && S.n_gran != 0u) ? 0u : 1u;
    const bool empty = have && S.n_gran == 0u;
    const uint32_t most = kTailsInTranscoder ? 0u : wave_max_u32(S.n_gran);     // (dec_transcode<0> has walked the tails: slot_rec is complete)
    for (uint32_t i = 0; i < most && lanes_where(t.found == 0u) != 0 && !(QOIMI_SCAN_EXP & 1); i += 4u) {              // four rows in flight (a lane's anchor is a few records back)
        u32x4 v[4];
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k) v[k] = S.granule(S.n_gran - 1u - i - k);           // (wrap past row 0: far beyond n_gran, granule() returns zeros)
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (i + k < S.n_gran) { tail_step(t, v[k].w, in.a_abs, in.ac); tail_step(t, v[k].z, in.a_abs, in.ac); tail_step(t, v[k].y, in.a_abs, in.ac); tail_step(t, v[k].x, in.a_abs, in.ac); }
    }
    if (empty) t.found = 0u;
    const SlotRec mine_r = kTailsInTranscoder ? in : tail_finish(t, in.a_abs, in.ac);
    const uint32_t mine = have ? slot_pack(mine_r) : kSlotIdentity;
    // ---- scan inside the workgroup
    uint32_t ipx = npix, isl = mine;
    wave_scan_px_slots(ipx, isl, lane);                                                      // inclusive over the wavefront
    if (lane == 63u) { s_wpx[wave] = ipx; s_wsl[wave] = isl; }
    __syncthreads();
    uint32_t wpx = 0u, wsl = kSlotIdentity, bpx = 0u, bsl = kSlotIdentity;                   // in front of this wavefront / the whole workgroup
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k) {
        const uint32_t a = s_wpx[k], b = s_wsl[k];
        if (k < wave) { wpx = scan_sat(wpx, a); wsl = slots_then(wsl, b); }
        bpx = scan_sat(bpx, a); bsl = slots_then(bsl, b);
    }
    // ---- look-back over the workgroups in front of this one in its image (first wavefront)
    if (wave == 0u) {
        u64* const st = p.scan_status;
        uint32_t epx = 0u, esl = kSlotIdentity;            // what stands in front of this workgroup: pixels, transfer from the image's start
        if (blk != 0u) {
            if (lane == 0) granule_store(&st[gb], scan_word(p, kScanAgg, bpx, bsl));
            // Every workgroup of a lone frame starts at about the same time, so the words in front of one are aggregates, not inclusive
            // prefixes, nearly all the way back: the look-back of the image's last workgroup is blk / 64 windows deep, and taken one
            // after the other each was a round trip through the L2 (20 us for this kernel on a 4K frame).  Up to kScanWin windows are
            // asked for together; a window that holds a word not yet written is asked for again.
            constexpr int kScanWin = 8;
            uint32_t look = blk;                           // the words of blocks [look, blk) are in (epx, esl)
            bool done = (QOIMI_SCAN_EXP & 2) != 0;
            while (!done) {
                u64 w[kScanWin];
#pragma unroll
                for (int k = 0; k < kScanWin; ++k) {
                    const int idx = (int)look - 64 * (k + 1) + (int)lane;                     // window k, lane 63: the nearest block not yet looked at
                    w[k] = idx >= 0 ? granule_load(&st[first_blk + (uint32_t)idx]) : scan_word(p, kScanIncl, 0u, kSlotIdentity);   // in front of the image: nothing
                }
#pragma unroll
                for (int k = 0; k < kScanWin; ++k) {
                    if (done) break;
                    const bool ok = ((uint32_t)(w[k] >> 46) & 0xFFFFu) == (p.epoch & 0xFFFFu) && (w[k] >> 62) != 0ull;
                    const u64 incl = lanes_where(ok && (w[k] >> 62) == 2ull);
                    const int stop = incl ? 63 - (int)__builtin_clzll(incl) : 0;              // nearest inclusive word (else: the whole window)
                    if (lanes_where(!ok && (int)lane >= stop) != 0ull) { __builtin_amdgcn_s_sleep(1); break; }   // a word this block must add is not there yet: its writer has started (ticket); ask again from this window on
                    const bool mine_w = (int)lane >= stop;
                    uint32_t vpx = mine_w ? (uint32_t)w[k] & kScanPxCap : 0u, vsl = mine_w ? scan_word_slots(w[k]) : kSlotIdentity;
                    wave_scan_px_slots(vpx, vsl, lane);
                    const uint32_t wp = read_lane(vpx, 63), ws = read_lane(vsl, 63);
                    epx = scan_sat(wp, epx); esl = slots_then(ws, esl);                        // the window first, then what was looked at before
                    if (incl) done = true;
                    else look -= 64u;                                                         // (no inclusive word among 64: all 64 were real blocks)
                }
            }
        }
        if (lane == 0) {
            granule_store(&st[gb], scan_word(p, kScanIncl, scan_sat(epx, bpx), slots_then(esl, bsl)));
            s_epx = epx; s_esl = esl;
        }
    }
    __syncthreads();
    // ---- every segment: pixels in front of it, speculated slot / alpha at its entry (S2: from the start pixel's hash and alpha)
    const uint32_t epx = s_epx, esl = s_esl;
    const uint32_t before = scan_sat(scan_sat(epx, wpx), from_lane_below(ipx, 0u));
    const uint32_t my_off = min(before, im.npx);
    uint32_t slot = hash_px(kInitPx), alpha = kInitPx >> 24;
    slot_apply(slot_unpack(slots_then(slots_then(esl, wsl), from_lane_below(isl, kSlotIdentity))), slot, alpha);
    if (have) { p.px_off[q] = my_off; p.slot_in[q] = (uint8_t)slot; p.alpha_in[q] = (uint8_t)alpha; p.slot_rec[q] = mine_r; }
    // n_active: segments that start before the pixel limit.  Offsets never fall, so only the wavefront that holds the last such segment reports.
    const u64 act = lanes_where(have && my_off < im.npx);
    if (act != 0ull) {
        const int top = 63 - (int)__builtin_clzll(act);
        const uint32_t j_top = q0 + wave * 64u + (uint32_t)top - im.seg_base;
        const uint32_t after = scan_sat(scan_sat(epx, wpx), read_lane(ipx, 63));              // pixels in front of the next wavefront's first segment
        const bool boundary = top < 63 || j_top + 1u >= im.nseg || after >= im.npx;
        if (boundary && lane == 0) atomicMax(&p.images[img].n_active, j_top + 1u);
    }
    if (threadIdx.x == 0 && q0 + kScanSegs >= im.seg_base + im.nseg) p.images[img].total_px = min(scan_sat(epx, bpx), im.npx);     // the image's last workgroup
}

// A wavefront's 64 x 64 table of dwords in the LDS ([row][lane], row stride 256 bytes) transposed IN PLACE: lane l swaps A[l + d][l]
// with A[l][l + d] for d = 1..31 (every lane) and d = 32 (lanes 0..31) - every pair once; an instruction's 64 accesses lie in 64
// different columns: no bank conflict.  What P3 / P4 keep per segment lives [slot][segment] in the LDS (a lane owns a column = a bank) and
// [segment][slot] in memory (a segment's 65 words in one piece): the lane-wise copy between the two was 65 instructions of 64 scattered
// 4 / 8-byte accesses 260 / 520 bytes apart - at 128-byte segments 7 us of dec_summarize_rec's 30 and 8 of dec_segments_rec's 39 on a
// lone 4K frame (profiles/r06_s4_state_io.txt).  Transposed, a row is a segment and leaves / arrives as one contiguous piece per instruction.
__device__ __forceinline__ void lds_transpose64(uint32_t base0, uint32_t lane) {       // base0: LDS byte address of A[0][0]
    __builtin_amdgcn_wave_barrier();
#pragma unroll 4
    for (uint32_t d = 1; d < 32u; ++d) {
        const uint32_t c = (lane + d) & 63u;
        lds_u32* const a = (lds_u32*)(base0 + c * 256u + lane * 4u);
        lds_u32* const b = (lds_u32*)(base0 + lane * 256u + c * 4u);
        const uint32_t x = *a, y = *b;
        *a = y; *b = x;
    }
    if (lane < 32u) {
        const uint32_t c = lane + 32u;
        lds_u32* const a = (lds_u32*)(base0 + c * 256u + lane * 4u);
        lds_u32* const b = (lds_u32*)(base0 + lane * 256u + c * 4u);
        const uint32_t x = *a, y = *b;
        *a = y; *b = x;
    }
    __builtin_amdgcn_wave_barrier();
}

// Source / mask codes of the symbolic table as BYTES, laid out so that a wavefront's access to 64 different rows is free
// of bank conflicts: the byte of (row r, lane l) sits in dword (r & 31) * 32 + (l & 31), byte 2 * (r >> 5) + (l >> 5) -
// the 32 lanes of either half hit 32 different banks whatever their rows.  ([row][lane] bytes put four lanes into one
// dword and a half-wavefront onto 16 banks: SQ_LDS_BANK_CONFLICT was 47 % of the LDS cycles of dec_summarize.)
__device__ __forceinline__ uint32_t symcode_addr(uint32_t lane_base, uint32_t row) {
    // lane_base = table base (4 KiB aligned) + (l & 31) * 4 + (l >> 5): bits 1 and 7..11 are clear, two bit-field inserts fill them
    const uint32_t a = ((row >> 4) & 2u) | (lane_base & ~2u);
    return ((row << 7) & 0xF80u) | (a & ~0xF80u);
}

// P3 on records.  One wavefront per 64 segments; 20 KiB of LDS (24 in the refinement rounds): eight per CU.
// Same step as dec_summarize / symr_step (qoi_decode_core.h).
// (a & m) | b as ONE instruction whatever else the compiler could share the parts with
__device__ __forceinline__ uint32_t and_or_b32(uint32_t a, uint32_t m, uint32_t b) {
    uint32_t r;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(m), "v"(b));
    return r;
}
// code byte of the plain form (dec_summarize_rec): source 0..64, bit 7 = r,g,b absolute -> the general form's code
// records of a PAIR (dec_transcode puts QOI_OP_RGB / QOI_OP_RGBA on even record indices): its first is class 2 or null, its second class 3 or null
__device__ __forceinline__ bool pair_head(uint32_t r) { return (r >> 30) == 2u || r == 0u; }
__device__ __forceinline__ bool pair_tail(uint32_t r) { return r >= 0xC0000000u || r == 0u; }
__device__ __forceinline__ uint32_t plain_code_general(uint32_t c) { return (c & 0x80u) ? (c & 0x7Fu) + kSymCodeRgb : c; }
template <bool REFINE>
__global__ __launch_bounds__(64) void dec_summarize_rec(DecParams p) {
    __shared__ __attribute__((aligned(16384))) uint32_t s_tabc[64 * 64];
    __shared__ __attribute__((aligned(4096))) uint8_t s_tabm[64 * 64];
    __shared__ uint8_t s_hint[REFINE ? 65 * 64 : 4];
    typedef __attribute__((address_space(3))) uint8_t lds_u8;
    const uint32_t lane = lane_id();
    if (p.tr_scan && *p.sync_fails != 0u) return;          // (a call whose first pass could not synchronise every segment is void up to here: the host takes the chains)
    if (REFINE && refine_pass_idle(p)) return;             // (the pass before this one changed nothing: a fixed point)
    if (p.s3_ctr && blockIdx.x == 0u)                      // the arrival counters of the state chain that follows this launch (dec_chain_state_l1q)
        for (uint32_t i = lane; i < p.n_images * kL2pWgs * (kL2Waves + 1u); i += 64u) p.s3_ctr[i] = 0u;
    const uint32_t q = blockIdx.x * 64u + lane;
    bool have = q < p.total_segs;
    const uint32_t img = find_image(p.images, p.n_images, have ? q : 0u);
    const DecImage im = p.images[img];
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    have = have && j >= im.start_seg && j < im.n_active;
    if (REFINE && p.only_flat) have = have && dec_image_is_flat(im.chunks_end, im.npx);      // the first round's extra passes
    if (!lanes_where(have)) return;
    RecSource S; S.init(p, blockIdx.x, lane, have ? p.rec_gran[q] : 0u);
    const uint32_t nblk = wave_max_u32((S.n_gran + 1u) >> 1);            // blocks of two granules = eight steps
    // Records are fetched kDepth blocks ahead.  A wavefront has 20 KiB of LDS, so eight of them share a CU and the record
    // stream (4 B per chunk, 2.7 x the QOI bytes of a photograph) has to be kept in flight by the few wavefronts there are:
    // one block ahead = 2 KiB per wavefront = 4 MiB over the chip, which at ~2 us of loaded HBM latency is 2 TB/s - the plain
    // form (900 cycles of work per block) would wait on every block.
#ifndef QOIMI_P3_DEPTH
#define QOIMI_P3_DEPTH 4
#endif
    constexpr uint32_t kDepth = QOIMI_P3_DEPTH;
    u32x4 ring[2u * kDepth];
#pragma unroll
    for (uint32_t i = 0; i < 2u * kDepth; ++i) ring[i] = S.granule_nt(i);
    const uint32_t tc_base = lds_addr_of(&s_tabc[lane]);                 // slot k at + k*256
    const uint32_t tm_lane = lds_addr_of(&s_tabm[0]) + (lane & 31u) * 4u + (lane >> 5);
    // PLAIN form, for as long as no lane of the wavefront has met a QOI_OP_RGBA record (opaque images: the whole segment): the
    // alpha of every value still is its source's, so a symbolic value is "entry word s + (dr,dg,db)" or "r,g,b with the alpha
    // of entry word s" and fits ONE dword - the code (s, or 128 + s) in the top byte, constants below.  The table is then a single LDS array, a step reads one word,
    // writes one word and tracks neither a code nor an alpha: 14 vector instructions instead of 39.  The first such record
    // converts table and pixel to the general form (constants + source/mask code, below) once, in place.
    // identity: slot k = entry slot k + 0, pixel = entry pixel + 0 (sym_init)
    bool plain = p.p3_plain != 0u;                                        // wave-uniform
    if (plain) { for (uint32_t k = 0; k < 64u; ++k) *(lds_u32*)(tc_base + k * 256u) = k << 24; }
    else { for (uint32_t k = 0; k < 64u; ++k) { *(lds_u32*)(tc_base + k * 256u) = 0u; *(lds_u8*)symcode_addr(tm_lane, k) = (uint8_t)k; } }
    uint32_t ppc = 64u << 24;                                             // plain running pixel: entry pixel + 0
    uint32_t pc = 0u, ph = 64u;                                           // general form: constants, code of the running pixel
    uint32_t slot, alpha, stash = 0u;
    if (REFINE) {
        const uint32_t* __restrict__ ent = p.entry + (size_t)(have ? q : 0u) * 65u;
        uint32_t epx = 0;
        if (have) {
            for (uint32_t k0 = 0; k0 < 64u; k0 += 16u) {
                uint32_t v[16];
#pragma unroll
                for (uint32_t k = 0; k < 16u; ++k) v[k] = ent[k0 + k];
#pragma unroll
                for (uint32_t k = 0; k < 16u; ++k) s_hint[(k0 + k) * 64u + lane] = (uint8_t)(v[k] >> 24);
            }
            epx = ent[64];
            s_hint[64u * 64u + lane] = (uint8_t)(epx >> 24);
        }
        slot = hash_px(epx); alpha = epx >> 24;
    } else {
        slot = have ? p.slot_in[q] : 0u; alpha = have ? p.alpha_in[q] : 0u;
    }
    const uint32_t alpha_in0 = alpha;
    // RUN chunks (and null records) store nothing new except as a stream's first chunk (SymState); the refinement rounds
    // skip the store by writing back the word they read (see dec_summarize)
    const bool skip_runs = REFINE && j != 0u;
    // read-ahead state of the plain form: address and word of the next step's table read, address and word of the last table write
    uint32_t ra_cur = and_or_b32(ring[0].x, 0x3F00u, tc_base), t_cur = 0u, wa_prev = ~0u, wv_prev = 0u;
    if (plain) t_cur = *(const lds_u32*)ra_cur;
    // One loop over the blocks, unrolled kDepth times (the ring is indexed statically: registers); a block of eight steps is
    // taken in the plain form or in the general one.  A block that holds a half of a QOI_OP_RGBA in any lane (class 3, or class 2
    // with the stash marker: exactly the records >= 0xBF000000) ends the plain form: the table is converted in front of it.
    // (no early exit from the unrolled blocks: the wavefront walks null records up to a multiple of kDepth blocks - with an exit
    // in the middle the compiler rotates the ring through register copies, and every copy waits for the load just issued)
    for (uint32_t blk0 = 0; blk0 < nblk; blk0 += kDepth) {
#pragma unroll
      for (uint32_t d = 0; d < kDepth; ++d) {
        const uint32_t blk = blk0 + d;
        const u32x4 c0 = ring[2u * d], c1 = ring[2u * d + 1u];
        const uint32_t rc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
        if (plain) {
            const uint32_t top = max(max(max(rc[0], rc[1]), max(rc[2], rc[3])), max(max(rc[4], rc[5]), max(rc[6], rc[7])));
            if (__builtin_expect(lanes_where(top >= 0xBF000000u) != 0, 0)) {
                // ---- plain -> general: split every table word into constants and code, the pixel likewise ----
                for (uint32_t k = 0; k < 64u; ++k) {
                    const uint32_t w = *(const lds_u32*)(tc_base + k * 256u);
                    *(lds_u32*)(tc_base + k * 256u) = w & 0x00FFFFFFu;
                    *(lds_u8*)symcode_addr(tm_lane, k) = (uint8_t)plain_code_general(w >> 24);
                }
                pc = ppc & 0x00FFFFFFu; ph = plain_code_general(ppc >> 24);
                slot &= 63u;
                // the alpha of the running pixel is its source's: the hinted one in a refinement round, else the entry alpha
                alpha = REFINE ? (uint32_t)s_hint[((ppc >> 24) & 0x7Fu) * 64u + lane] : alpha_in0;
                plain = false;
            }
        }
        if (plain) {
            // Two wavefronts per SIMD (the tables fill the LDS) issue one vector instruction every ~10 cycles each, whatever the
            // instruction, as long as the next one does not depend on it (tools/ubench/valu_tput.hip: 6.9 ticks of 1.5 cycles): a
            // step costs its VECTOR instruction count, scalar instructions ride along.  So
            //  * there is no branch inside a block of eight steps (a branch holds a wavefront for ~40 cycles even when it is not
            //    taken): the block is either free of QOI_OP_RGB in every lane or takes the body that handles it with selects;
            //  * the table word an INDEX names is read ONE STEP AHEAD of its use - before the previous step's table write, which
            //    is forwarded from registers when the addresses match - so the LDS round trip is not part of a step's chain.
            const u32x4 nx = ring[2u * ((d + 1u) % kDepth)];                // first record of the next block (null past the end)
            const uint32_t rx[9] = {rc[0], rc[1], rc[2], rc[3], rc[4], rc[5], rc[6], rc[7], nx.x};
            // whether the block has a QOI_OP_RGB in some lane: the sign bit of the OR of its records
            const uint32_t any = rc[0] | rc[1] | rc[2] | rc[3] | rc[4] | rc[5] | rc[6] | rc[7];
            auto plain_block = [&](auto rgb_tag) {
                constexpr bool RGB = decltype(rgb_tag)::value;
#pragma unroll
                for (uint32_t u = 0; u < 8u; ++u) {
                    const uint32_t rec = rx[u];
                    const uint32_t ra_next = and_or_b32(rx[u + 1u], 0x3F00u, tc_base);      // INDEX records carry the slot in bits 8..13 too
                    const uint32_t t_next = *(const lds_u32*)ra_next;                  // read one step ahead, i.e. before this step's write
                    const uint32_t t = ra_cur == wa_prev ? wv_prev : t_cur;            // ... so the previous step's write is forwarded here
                    const bool isabs = rec >= 0x40000000u;                            // INDEX (or RGB: the sign bit)
                    add_byte0(ppc, rec); add_byte1(ppc, rec); add_byte2(ppc, rec);
                    const uint32_t s_rel = __builtin_amdgcn_udot4(rec, 0x00070503u, slot, false);
                    const uint32_t code = ppc;                                        // the code byte is not touched by the byte adds
                    ppc = isabs ? t : ppc;
                    slot = isabs ? rec : s_rel;                                       // modulo 64: masked where it addresses the table
                    if (RGB) {
                        // QOI_OP_RGB keeps the plain form: r,g,b become absolute (bit 7 of the code), the alpha stays the source's
                        const bool hi = (int32_t)rec < 0;
                        const uint32_t rgbc = (rec & 0x00FFFFFFu) | 0x80000000u;
                        const uint32_t a_src = REFINE ? (uint32_t)s_hint[((code >> 24) & 0x7Fu) * 64u + lane] : alpha_in0;
                        ppc = hi ? ((code & 0x7F000000u) | rgbc) : ppc;
                        slot = hi ? __builtin_amdgcn_udot4(rgbc, 0x00070503u, 11u * a_src, false) : slot;
                    }
                    uint32_t waddr = and_or_b32(slot << 8, 0x3F00u, tc_base), wval = ppc;    // index update after every chunk (qoi.h:577)
                    if (REFINE) {
                        const bool keep = skip_runs && (rec & 0xC0FFFFFFu) == 0u;       // a RUN / null record read slot 0 (payload 0): it goes back
                        waddr = keep ? tc_base : waddr; wval = keep ? t : ppc;
                    }
                    *(lds_u32*)waddr = wval;
                    wa_prev = waddr; wv_prev = wval; ra_cur = ra_next; t_cur = t_next;
                }
            };
            if (lanes_where((int32_t)any < 0) == 0) plain_block(std::false_type{}); else plain_block(std::true_type{});
        } else if (!REFINE && lanes_where(!(pair_head(rc[0]) && pair_tail(rc[1]) && pair_head(rc[2]) && pair_tail(rc[3]) &&
                                            pair_head(rc[4]) && pair_tail(rc[5]) && pair_head(rc[6]) && pair_tail(rc[7]))) == 0) {
            // ---- a block of PAIRS in every lane: (stash half, alpha half) = QOI_OP_RGBA, (record, null) = QOI_OP_RGB, (null, null) -
            // RGBA-dense content (noise, sprites with many alpha levels); the transcoder puts these chunks on even record indices
            // for this.  Such a chunk names r,g,b - and with QOI_OP_RGBA the alpha - absolutely (qoi.h:547-557): nothing is read from
            // the table, a pair is a dozen vector instructions instead of two passes through the rare body below.  Same function
            // of the records as those: RGBA sets pc = r,g,b | a << 24, code = all absolute, alpha = a; RGB sets r,g,b and keeps alpha
            // and its code; both name the slot QOI_COLOR_HASH; a null pair stores the running word where it already stands.
#pragma unroll
            for (uint32_t u = 0; u < 8u; u += 2u) {
                const uint32_t rs = rc[u], ra = rc[u + 1u];
                const bool real = rs != 0u, is_a = ra != 0u;
                const uint32_t npc = (rs & 0x00FFFFFFu) | (is_a ? (ra << 24) : (pc & 0xFF000000u));
                const uint32_t hb = ph < kSymCodeRgb ? ph + kSymCodeRgb : ph;
                ph = is_a ? kSymCodeAbs : (real ? hb : ph);
                alpha = is_a ? (ra & 0xFFu) : alpha;
                pc = real ? npc : pc;
                slot = (real ? __builtin_amdgcn_udot4(pc, 0x00070503u, 0u, false) + 11u * alpha : slot) & 63u;
                *(lds_u32*)(tc_base + (slot << 8)) = pc;                   // index update after every chunk (qoi.h:577)
                *(lds_u8*)symcode_addr(tm_lane, slot) = (uint8_t)ph;
            }
        } else {
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                const uint32_t rec = rc[u];
                const uint32_t idx = rec & 63u;
                const uint32_t tm_rd = symcode_addr(tm_lane, idx);
                const uint32_t t_c = *(const lds_u32*)(tc_base + (idx << 8));
                const uint32_t t_m = *(const lds_u8*)tm_rd;
                uint32_t pc_rel = pc;
                add_byte0(pc_rel, rec); add_byte1(pc_rel, rec); add_byte2(pc_rel, rec);
                const bool hi = (int32_t)rec < 0, lo = (int32_t)(rec << 1) < 0;
                const uint32_t s_rel = slot + __builtin_amdgcn_udot4(rec, 0x00070503u, 0u, false);
                // alpha an INDEX chunk leaves: the named entry's (hinted where it is still symbolic)
                const uint32_t th = REFINE ? (uint32_t)s_hint[(t_m < kSymCodeRgb ? t_m : (t_m - kSymCodeRgb) & 0x7Fu) * 64u + lane] : alpha_in0;
                const uint32_t ta = t_m == kSymCodeAbs ? (t_c >> 24) : th;
                // two complete bodies: the common one knows nothing of QOI_OP_RGB / QOI_OP_RGBA (no selects on `hi`, no write-back)
                if (__builtin_expect(lanes_where(hi) != 0, 0)) {
                    const uint32_t rgb = rec & 0x00FFFFFFu;
                    const bool is_stash = hi && !lo && ((rec >> 24) & 63u) == kRecStash;
                    const uint32_t a_new = rec & 0xFFu;
                    const uint32_t pc_rgb = (pc & 0xFF000000u) | rgb, pc_abs = stash | (rec << 24);
                    const uint32_t l_rgb = __builtin_amdgcn_udot4(rgb, 0x00070503u, 0u, false), l_st = __builtin_amdgcn_udot4(stash, 0x00070503u, 0u, false);
                    const uint32_t sb = lo ? l_st + 11u * a_new : l_rgb + 11u * alpha;
                    const uint32_t pb = lo ? pc_abs : pc_rgb;
                    const uint32_t hb = lo ? kSymCodeAbs : (ph < kSymCodeRgb ? ph + kSymCodeRgb : ph);
                    const uint32_t ab = lo ? a_new : alpha;
                    const uint32_t pa = lo ? t_c : pc_rel, ha = lo ? t_m : ph, sa = lo ? idx : s_rel, aa = lo ? ta : alpha;
                    const bool keep = is_stash || (skip_runs && (rec & 0xC0FFFFFFu) == 0u);   // the table stays as it is: the word read goes back
                    if (!is_stash) {
                        pc = hi ? pb : pa; ph = hi ? hb : ha; slot = (hi ? sb : sa) & 63u; alpha = hi ? ab : aa;
                    } else {
                        stash = rgb;
                    }
                    const uint32_t wslot = keep ? idx : slot, wc = keep ? t_c : pc, wm = keep ? t_m : ph;
                    *(lds_u32*)(tc_base + (wslot << 8)) = wc;          // index update after every chunk (qoi.h:577)
                    *(lds_u8*)symcode_addr(tm_lane, wslot) = (uint8_t)wm;
                } else {
                    pc = lo ? t_c : pc_rel; ph = lo ? t_m : ph; slot = (lo ? idx : s_rel) & 63u; alpha = lo ? ta : alpha;
                    if (REFINE) {
                        const bool keep = skip_runs && (rec & 0xC0FFFFFFu) == 0u;
                        const uint32_t wslot = keep ? idx : slot, wc = keep ? t_c : pc, wm = keep ? t_m : ph;
                        *(lds_u32*)(tc_base + (wslot << 8)) = wc;
                        *(lds_u8*)symcode_addr(tm_lane, wslot) = (uint8_t)wm;
                    } else {
                        *(lds_u32*)(tc_base + (slot << 8)) = pc;       // index update after every chunk (qoi.h:577)
                        *(lds_u8*)symcode_addr(tm_lane, slot) = (uint8_t)ph;
                    }
                }

            }
        }
        // the ring slot is refilled when the block is through with it: the load goes straight into the registers the block
        // after next ... reads (a refill at the top of the block lands in shadow registers and the copies wait for it)
        ring[2u * d] = S.granule_nt(2u * (blk + kDepth)); ring[2u * d + 1u] = S.granule_nt(2u * (blk + kDepth) + 1u);
      }
    }
    if (plain) {
        // never left the plain form (code and constants share the word): the table is turned in the LDS and leaves segment by segment,
        // 512 contiguous bytes per instruction (lds_transpose64)
        const u64 hv = lanes_where(have);
        const uint32_t base0 = tc_base - lane * 4u;
        lds_transpose64(base0, lane);
        sym_t* const blk = p.summary + (size_t)blockIdx.x * 64u * 65u;
        for (uint32_t sgm = 0; sgm < 64u; ++sgm) {
            if (!((hv >> sgm) & 1ull)) continue;
            const uint32_t w = *(const lds_u32*)(base0 + sgm * 256u + lane * 4u);
            blk[(size_t)sgm * 65u + lane] = (sym_t)(w & 0x00FFFFFFu) | ((sym_t)sym_code_expand(plain_code_general(w >> 24)) << 32);
        }
        if (have) blk[(size_t)lane * 65u + 64u] = (sym_t)(ppc & 0x00FFFFFFu) | ((sym_t)sym_code_expand(plain_code_general(ppc >> 24)) << 32);
    } else if (have) {
        sym_t* dst = p.summary + (size_t)q * 65u;
        for (uint32_t k = 0; k < 64u; ++k)
            dst[k] = (sym_t)*(const lds_u32*)(tc_base + k * 256u) | ((sym_t)sym_code_expand(*(const lds_u8*)symcode_addr(tm_lane, k)) << 32);
        dst[64] = (sym_t)pc | ((sym_t)sym_code_expand(ph) << 32);
    }
}

// Pixel sink of dec_segments_rec: LaneWriter's ring (32 pixels per lane) with a BRANCH-FREE drain.  Once per block of
// eight steps every lane that holds a complete, 16-pixel aligned group writes it as four (OCH 3: three) back-to-back
// 16-byte buffer stores; lanes without one aim outside the buffer descriptor and the hardware drops their part.  The
// instructions are issued whatever the lanes hold, so their number is static - and that is the point: vector memory
// operations retire in order and a wavefront has one counter for loads and stores (gfx9 family), so the wait for the
// next block's records is `s_waitcnt vmcnt(4)` - records yes, this block's stores no.  With the stores under
// data-dependent loops (LaneWriter::drain) the compiler has to wait with vmcnt(0): profiles/r02 - 75 % of the wavefront
// cycles of the first dec_segments_rec were spent there, waiting for store acknowledgements.
// Group stores go through a descriptor based at the image of the wavefront's first segment (32-bit offsets); a lane whose
// image lies 2 GiB or more behind that base (never with sane strides) uses plain stores on a rare path.
template <int OCH, uint32_t RING_ = 32, uint32_t GROUP_ = 16>
struct BurstWriter : LaneWriter<OCH, RING_, GROUP_> {
    typedef LaneWriter<OCH, RING_, GROUP_> Base;
    static constexpr uint32_t kG = GROUP_;
    static_assert(RING_ == 2u * GROUP_ && (GROUP_ == 16u || GROUP_ == 8u), "a ring of two groups");
    __amdgpu_buffer_rsrc_t rs;
    uint32_t boff;         // byte offset of the lane's image in the descriptor; kFar: not addressable through it
    static constexpr uint32_t kFar = 0xFFFFFFFFu, kNowhere = 0x7FFFFFF0u, kRange = 0x7FFF0000u;
    __device__ __forceinline__ void init(uint32_t row_addr, uint8_t* wave_base, uint8_t* image, uint32_t image_bytes, uint32_t px_pos) {
        Base::init(row_addr, image, px_pos);
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)wave_base, 0, (int)kRange, 0x00020000);
        const unsigned long long d = (unsigned long long)(image - wave_base);
        boff = d + image_bytes < (unsigned long long)kRange ? (uint32_t)d : kFar;
    }
    __device__ __forceinline__ void drain_block() {
        // head of a segment / after a long run: up to the next group boundary pixel by pixel (LaneWriter::drain's first loop)
        if (__builtin_expect(lanes_where((this->fpos & (kG - 1u)) != 0u && this->fpos < this->ppos) != 0, 0)) {
            while ((this->fpos & (kG - 1u)) != 0u && this->fpos < this->ppos) {
                if ((this->fpos & 3u) == 0u && this->fpos + 4u <= this->ppos) { this->store4(this->fpos, this->at(this->fpos), this->at(this->fpos + 1u), this->at(this->fpos + 2u), this->at(this->fpos + 3u)); this->fpos += 4u; }
                else { this->store_one(this->fpos, this->at(this->fpos)); ++this->fpos; }
            }
        }
        const bool ready = (this->fpos & (kG - 1u)) == 0u && this->ppos - this->fpos >= kG;
        if (__builtin_expect(lanes_where(ready && boff == kFar) != 0, 0)) {           // image out of the descriptor's reach
            if (ready && boff == kFar) Base::drain();
        }
        const bool go = ready && boff != kFar;
        if (OCH == 4) {
            // Cooperative: the bytes of an owner's group are written by kG/4 ADJACENT LANES, 16 bytes each (lane t: piece
            // t % (kG/4) of owner t / (kG/4) + (256/kG) j in instruction j) - contiguous 4 kG-byte pieces instead of 64 scattered
            // 16-byte ones.  The address path of the CU was the limit with one piece per lane (~230 CU-cycles per store
            // instruction, SQ_WAIT_INST_ANY 43 % of the wavefront cycles, profiles/r02).
            constexpr uint32_t kLpo = kG / 4u, kOwners = 64u / kLpo;                // lanes per owner, owners per instruction
            const uint32_t lane = (this->row >> 2) & 63u;
            // the group's byte offset in the descriptor (below 2^31: kRange) with "the group is the upper half of the ring" on top:
            // ONE cross-lane gather per owner instead of two
            const uint32_t A = (go ? boff + this->fpos * 4u : kNowhere) | ((this->fpos & kG) ? 0x80000000u : 0u);
            const uint32_t piece = lane & (kLpo - 1u);
#pragma unroll
            for (uint32_t j = 0; j < kLpo; ++j) {
                const uint32_t owner = lane / kLpo + kOwners * j;
                const uint32_t Ag = gather_lane(A, owner);
                const uint32_t Ao = Ag & 0x7FFFFFFFu, rbo = (Ag >> 31) * kG;
                const uint32_t raddr = this->row - lane * 4u + owner * 4u + ((rbo + 4u * piece) << 8);
                u32x4 w;
                w.x = *(const lds_u32*)(raddr); w.y = *(const lds_u32*)(raddr + 256u); w.z = *(const lds_u32*)(raddr + 512u); w.w = *(const lds_u32*)(raddr + 768u);
                __builtin_amdgcn_raw_buffer_store_b128(w, rs, Ao + 16u * piece, 0, QOIMI_P4_STORE_AUX);    // kNowhere + 48 is still outside
            }
        } else if (kG == 16u) {
            // 3-byte pixels, the same way in two stages: a lane with a complete group packs its 16 pixels into 12 dwords IN PLACE
            // (the group's ring slots are free once they are read), then three of every four adjacent lanes write one owner's 48
            // bytes as 16-byte pieces.  (One lane per group with three stores of its own was 1.76 x the time of the 4-byte path
            // while writing 25 % fewer bytes: 64 scattered 16-byte pieces per store instruction.)
            const uint32_t rbase = this->row + ((this->fpos & kG) << 8);
            if (go) {
                uint32_t v[kG];
#pragma unroll
                for (uint32_t k = 0; k < kG; ++k) v[k] = *(const lds_u32*)(rbase + k * 256u);
#pragma unroll
                for (uint32_t k = 0; k < kG; k += 4u) {
                    const uint32_t a = v[k] & 0xFFFFFFu, b = v[k + 1u] & 0xFFFFFFu, c = v[k + 2u] & 0xFFFFFFu, e = v[k + 3u] & 0xFFFFFFu;
                    const uint32_t o = rbase + 3u * (k >> 2) * 256u;
                    *(lds_u32*)(o) = a | (b << 24); *(lds_u32*)(o + 256u) = (b >> 8) | (c << 16); *(lds_u32*)(o + 512u) = (c >> 16) | (e << 8);
                }
            }
            const uint32_t lane = (this->row >> 2) & 63u;
            const uint32_t A = (go ? boff + this->fpos * 3u : kNowhere) | ((this->fpos & kG) ? 0x80000000u : 0u);   // (as above)
            const uint32_t piece = lane & 3u;                                      // piece 3 does not exist: 48 bytes per group
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                const uint32_t owner = lane / 4u + 16u * j;
                const uint32_t Ag = gather_lane(A, owner);
                const uint32_t Ao = Ag & 0x7FFFFFFFu, rbo = (Ag >> 31) * kG;
                const uint32_t raddr = this->row - lane * 4u + owner * 4u + ((rbo + 4u * piece) << 8);
                u32x4 w;
                w.x = *(const lds_u32*)(raddr); w.y = *(const lds_u32*)(raddr + 256u); w.z = *(const lds_u32*)(raddr + 512u); w.w = *(const lds_u32*)(raddr + 768u);
                __builtin_amdgcn_raw_buffer_store_b128(w, rs, piece < 3u ? Ao + 16u * piece : kNowhere, 0, QOIMI_P4_STORE_AUX);
            }
        } else {
            const uint32_t rbase = this->row + ((this->fpos & kG) << 8);
            uint32_t v[kG];
#pragma unroll
            for (uint32_t k = 0; k < kG; ++k) v[k] = *(const lds_u32*)(rbase + k * 256u);
            const uint32_t off = go ? boff + this->fpos * (uint32_t)OCH : kNowhere;
            uint32_t d[3u * kG / 4u];                                             // kG pixels -> 3 kG / 4 dwords of packed r,g,b
#pragma unroll
            for (uint32_t k = 0; k < kG; k += 4u) {
                const uint32_t a = v[k] & 0xFFFFFFu, b = v[k + 1u] & 0xFFFFFFu, c = v[k + 2u] & 0xFFFFFFu, e = v[k + 3u] & 0xFFFFFFu;
                d[3u * (k >> 2)] = a | (b << 24); d[3u * (k >> 2) + 1u] = (b >> 8) | (c << 16); d[3u * (k >> 2) + 2u] = (c >> 16) | (e << 8);
            }
            if (kG == 16u) {
#pragma unroll
                for (uint32_t k = 0; k < 12u; k += 4u) {
                    u32x4 w; w.x = d[k]; w.y = d[k + 1u]; w.z = d[k + 2u]; w.w = d[k + 3u];
                    __builtin_amdgcn_raw_buffer_store_b128(w, rs, off + 4u * k, 0, QOIMI_P4_STORE_AUX);
                }
            } else {                                                               // 24 bytes: 16 + 8
                u32x4 w; w.x = d[0]; w.y = d[1]; w.z = d[2]; w.w = d[3];
                __builtin_amdgcn_raw_buffer_store_b128(w, rs, off, 0, QOIMI_P4_STORE_AUX);
                typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
                u32x2 w2; w2.x = d[4]; w2.y = d[5];
                __builtin_amdgcn_raw_buffer_store_b64(w2, rs, off + 16u, 0, QOIMI_P4_STORE_AUX);
            }
        }
        this->fpos += go ? kG : 0u;
    }
};

// P4 on records: genuine decode of every active segment + exit-state check (lane = segment), see dec_segments.
// 16 KiB of colour tables + the 4 KiB pixel ring = 20 KiB: eight wavefronts per CU.
// FLAT (round 5): the segments of "flat" images (dec_image_is_flat: UI frames, constant frames - their stream is a fraction of a
// byte per pixel, nearly all of it QOI_OP_RUN) run as a second launch of this kernel that writes almost no pixels itself: P4 wrote a
// run lane by lane in 16-byte pieces, 64 lanes 0.3-30 KB apart (34 GB in 19 ms on 1024 UI frames, a third of what the memory system
// gives coalesced writes).  Here a lane keeps a SPAN - the pixels since its pixel value last changed: a chunk that names a pixel and
// every QOI_OP_RUN behind it (a run is cut every 62 pixels, qoi.h:417; the UI tile's first pixel and the 95 repeats behind it are one
// span).  A span that ends with eight pixels or more becomes a 16-byte descriptor (first pixel, pixels, value) in the segment's
// descriptor region and dec_expand_runs writes all descriptors of the launch afterwards with whole wavefronts, 1 KiB per store
// instruction; shorter ones (a flat image holds few: an eighth of its pixels at most are chunks) are stored by the lane.  No pixel ring.
// Segments that left descriptors queue up (run_queue) for the expander.
// (First form of the round: only the 4-pixel aligned middle of a run of twelve or more as a descriptor, head and tail through the
// ring, the ring written out in front of every run - six scattered stores of 4-16 bytes per UI tile row: 8 ms per 1024 frames.)
template <int OCH, bool FLAT>
__global__ __launch_bounds__(64) void dec_segments_rec(DecParams p) {
#ifndef QOIMI_SEGREC_GROUP
#define QOIMI_SEGREC_GROUP 16
#endif
    typedef BurstWriter<OCH, 2 * QOIMI_SEGREC_GROUP, QOIMI_SEGREC_GROUP> Writer;
    constexpr uint32_t kTabDw = 64u * 64u, kOutDw = FLAT ? 64u : Writer::kRing * 64u;        // (FLAT: no pixel ring)
    constexpr uint32_t kDrainEvery = Writer::kGroup / 2u;      // steps between drains: they add <= 2 pixels each to what a drain leaves (< one group)
#ifndef QOIMI_SEGREC_PAD_DW
#define QOIMI_SEGREC_PAD_DW 0
#endif
    __shared__ __attribute__((aligned(16384))) uint32_t s_mem[kTabDw + kOutDw + QOIMI_SEGREC_PAD_DW];
    uint32_t* const s_tab = s_mem;
    uint32_t* const s_out = s_mem + kTabDw;
    const uint32_t lane = lane_id();
    if (p.tr_scan && *p.sync_fails != 0u) return;
    if (QOIMI_SEGREC_PAD_DW && p.total_segs == 0xFFFFFFFFu) s_mem[kTabDw + kOutDw + lane] = 0u;     // keeps the padding (occupancy experiments)
    const uint32_t q = blockIdx.x * 64u + lane;
    bool have = q < p.total_segs;
    const uint32_t img = find_image(p.images, p.n_images, have ? q : 0u);
    const DecImage im = p.images[img];
    const uint32_t j = (have ? q : im.seg_base) - im.seg_base;
    have = have && j >= im.start_seg && j < im.n_active && (im.desc_base != kNoRunDesc) == FLAT;
    if (!lanes_where(have)) return;
    const uint32_t limit = im.npx;
    const uint32_t px_first = have ? p.px_off[q] : 0u;
    // FLAT: the lane's run descriptors (8 bytes each: start pixel, pixels) and the run in the making
    // (the other images, desc_all: QOI_OP_RUNs of twelve pixels or more, and the runs that follow them directly, leave a descriptor too -
    // in the segment's slot of the symbolic summaries, which are dead once the entry states stand: kSummaryDescs of them)
    uint4* const my_desc = FLAT ? p.run_desc + (size_t)(im.desc_base + j) * p.desc_cap : reinterpret_cast<uint4*>(p.summary + (size_t)(have ? q : 0u) * 65u);
    uint32_t n_desc = 0u, span_start = px_first, span_len = 0u;     // FLAT: the pixels since the value last changed: [span_start, span_start + span_len), all of them px
    uint32_t span_px = 0u;                                          // the other images: the span of long runs in the making holds this pixel
    uint32_t n_long = 0u;                                           // QOI_OP_RUNs of kLongRun pixels or more this lane met (calls of a few images: what the next call's desc_all goes by)
    RecSource S; S.init(p, blockIdx.x, lane, have && px_first < limit ? p.rec_gran[q] : 0u);
    const uint32_t nblk = wave_max_u32((S.n_gran + 1u) >> 1);
#ifndef QOIMI_P4_DEPTH
#define QOIMI_P4_DEPTH 2
#endif
    constexpr uint32_t kDepth = QOIMI_P4_DEPTH;                  // blocks of records in flight (see dec_summarize_rec)
    u32x4 ring[2u * kDepth];
#pragma unroll
    for (uint32_t i = 0; i < 2u * kDepth; ++i) ring[i] = S.granule(i);
    Writer W;
    {   // descriptor base: the image of the wavefront's first segment (its lanes' images follow it in memory)
        const uint32_t q0 = blockIdx.x * 64u;
        const uint32_t img0 = find_image(p.images, p.n_images, q0 < p.total_segs ? q0 : 0u);
        W.init(lds_addr_of(&s_out[lane]), p.pixels + (size_t)p.images[img0].out_index * p.pixel_stride, p.pixels + (size_t)im.out_index * p.pixel_stride,
               im.npx * (uint32_t)OCH, px_first);
    }
    LdsTab32 tab{&s_tab[lane]};
    const uint32_t tab_base = lds_addr_of(&s_tab[lane]);
    const uint32_t* __restrict__ ent = p.entry + (size_t)(have ? q : 0u) * 65u;
    uint32_t px = 0, stash = 0;
    if (have) {
        // (read as whole segments in contiguous pieces and turned in the LDS - lds_transpose64, as dec_summarize_rec writes its summaries -
        // this kernel got SLOWER, 39.2 -> 44.4 us on a lone 4K frame: the two turns cost more than the 65 + 65 scattered loads, profiles/r06_s5)
        for (uint32_t k0 = 0; k0 < 64u; k0 += 16u) {          // 16 loads in flight, then 16 LDS writes
            uint32_t v[16];
#pragma unroll
            for (uint32_t k = 0; k < 16u; ++k) v[k] = ent[k0 + k];
#pragma unroll
            for (uint32_t k = 0; k < 16u; ++k) tab.set(k0 + k, v[k]);
        }
        px = ent[64];
    }
    constexpr uint32_t kLongRun = 12;
    // only a wavefront that holds a segment which may reach the image's pixel limit pays for the clipping (see dec_segments)
    const bool clip_lane = have && (j + 1u >= im.n_active || p.px_off[q + 1u] >= limit);
    // FLAT: the lane's span ends (its pixels have the value v): a descriptor for the expander, or - a few pixels - stored here
    constexpr uint32_t kMinSpan = 8;
    // (a descriptor is written and read as two 8-byte halves: the slots in the summary region lie 520 bytes apart - 8-byte aligned, no more)
    auto put_desc = [&](uint32_t at, uint32_t start, uint32_t len, uint32_t v) {
        uint2* d = reinterpret_cast<uint2*>(my_desc + at);
        d[0] = make_uint2(start, len); d[1] = make_uint2(v, 0u);
    };
    auto close_span = [&](uint32_t v) {
        if (span_len >= kMinSpan && n_desc < p.desc_cap) { put_desc(n_desc, span_start, span_len, v); ++n_desc; }
        else for (uint32_t i = 0; i < span_len; ++i) W.store_one(span_start + i, v);       // (no room: never - a span of eight takes two records at least)
        span_len = 0u;
    };
    auto run = [&](auto clip_tag) {
        constexpr bool CLIP = decltype(clip_tag)::value;
        // (no early exit from the unrolled blocks and the ring slot refilled at the END of its block: see dec_summarize_rec)
        // one block of eight steps on ring slot D (a compile-time index: `#pragma unroll` over the slots was refused for the
        // 3-channel instantiation, the ring went to scratch memory and every block waited for its own loads - 1.8 x the time)
        auto block = [&](auto dtag, const uint32_t blk) {
            constexpr uint32_t d = decltype(dtag)::value;
            const u32x4 c0 = ring[2u * d], c1 = ring[2u * d + 1u];      // the loads issued kDepth - 1 blocks (of eight steps) ago
            if (!FLAT) W.drain_block();                         // a static number of stores (BurstWriter)
            const uint32_t rc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
            // Which steps of the block have a QOI_OP_RGB / QOI_OP_RGBA record, and which a run of three pixels or more, in some lane:
            // sixteen compares up front, SCALAR tests at the steps.  (A branch on a vector compare made in the step holds the
            // wavefront for ~40 cycles even when it is not taken, tools/ubench/valu_lat.hip - there were two per step.)
            unsigned long long hm[8], lm[8];
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) { hm[u] = __ballot((int32_t)rc[u] < 0); lm[u] = __ballot((rc[u] & 0x3F000000u) > 0x02000000u); }
            // A block of PAIRS in every lane - (stash half, alpha half) = QOI_OP_RGBA, (record, null) = QOI_OP_RGB, (null, null) - is
            // RGBA-dense content (see the same test in dec_summarize_rec): the pair names r,g,b and perhaps alpha absolutely
            // (qoi.h:547-557), one short step per pair instead of two passes through the rare body.  Looked for only where the
            // block's first record is such a chunk in some lane, so photographs never pay for the question.
            if (!FLAT && __builtin_expect(hm[0] != 0ull, 0) &&
                lanes_where(!(pair_head(rc[0]) && pair_tail(rc[1]) && pair_head(rc[2]) && pair_tail(rc[3]) &&
                              pair_head(rc[4]) && pair_tail(rc[5]) && pair_head(rc[6]) && pair_tail(rc[7]))) == 0) {
#pragma unroll
                for (uint32_t u = 0; u < 8u; u += 2u) {
                    const uint32_t rs = rc[u], ra = rc[u + 1u];
                    bool real = rs != 0u;
                    if (CLIP) real = real && W.ppos < limit;              // at the pixel limit the decoder has stopped (qoi.h:540)
                    const uint32_t npx = (rs & 0x00FFFFFFu) | (ra != 0u ? (ra << 24) : (px & 0xFF000000u));
                    px = real ? npx : px;
                    const uint32_t h = __builtin_amdgcn_udot4(px, 0x0B070503u, 0u, false);
                    *(lds_u32*)(((h << 8) & 0x3F00u) | tab_base) = px;    // qoi.h:577
                    W.put2n(px, real ? 1u : 0u);
                }
                ring[2u * d] = S.granule(2u * (blk + kDepth)); ring[2u * d + 1u] = S.granule(2u * (blk + kDepth) + 1u);
                return;
            }
#pragma unroll
            for (uint32_t u = 0; u < 8u; ++u) {
                if (!FLAT && kDrainEvery < 8u && u == 4u) W.drain_block();
                const uint32_t rec = rc[u];
                uint32_t rem = (rec >> 24) & 63u;                                 // 0: null record
                const uint32_t px_before = px;
                // two complete bodies (see dec_summarize_rec: a step costs its instruction count): the common one knows nothing of
                // QOI_OP_RGB / QOI_OP_RGBA
                if (__builtin_expect(hm[u] != 0ull, 0)) {                        // QOI_OP_RGB keeps the alpha; QOI_OP_RGBA = stash record + alpha record (qoi.h:548-557)
                    const uint32_t idx8 = rec & 0x3F00u;
                    const uint32_t t = *(const lds_u32*)(tab_base + idx8);
                    uint32_t rel = px;
                    add_byte0(rel, rec); add_byte1(rel, rec); add_byte2(rel, rec);
                    const bool hi = (int32_t)rec < 0, lo = (int32_t)(rec << 1) < 0;
                    const uint32_t a = lo ? t : rel;
                    const uint32_t rgb = rec & 0x00FFFFFFu;
                    const bool is_stash = hi && !lo && rem == kRecStash;
                    const uint32_t b = lo ? (stash | (rec << 24)) : ((px & 0xFF000000u) | rgb);
                    uint32_t npxl = is_stash ? px : (hi ? b : a);
                    stash = is_stash ? rgb : stash;
                    rem = is_stash ? 0u : rem;
                    if (CLIP) npxl = W.ppos < limit ? npxl : px;         // at the pixel limit the decoder has stopped (qoi.h:540)
                    px = npxl;
                    // index[QOI_COLOR_HASH(px) % 64] = px after every chunk (qoi.h:577); a stash record puts back what it read
                    const uint32_t h = __builtin_amdgcn_udot4(px, 0x0B070503u, 0u, false);
                    const uint32_t waddr = is_stash ? idx8 : ((h & 63u) << 8), wval = is_stash ? t : px;
                    *(lds_u32*)(tab_base + waddr) = wval;
                } else {
                    const uint32_t t = *(const lds_u32*)((rec & 0x3F00u) | tab_base);   // slot an INDEX names: bits 8..13 of its record
                    const bool lo = rec >= 0x40000000u;                      // INDEX (no lane has bit 31 here)
                    if (CLIP) {
                        uint32_t rel = px;
                        add_byte0(rel, rec); add_byte1(rel, rec); add_byte2(rel, rec);
                        const uint32_t npxl = lo ? t : rel;
                        px = W.ppos < limit ? npxl : px;
                    } else {
                        add_byte0(px, rec); add_byte1(px, rec); add_byte2(px, rec);
                        px = lo ? t : px;
                    }
                    const uint32_t h = __builtin_amdgcn_udot4(px, 0x0B070503u, 0u, false);
                    *(lds_u32*)(((h << 8) & 0x3F00u) | tab_base) = px;       // qoi.h:577
                }
                if (CLIP) rem = min(rem, limit - W.ppos);                 // over-long run clipped (Appendix B item 8)
                if (FLAT) {
                    // a record that leaves the pixel as it is (QOI_OP_RUN, a null record) lengthens the span; one that names a pixel ends it
                    // and opens the next (a stash half yields no pixel: rem is 0)
                    const bool names = (rec & 0xC0FFFFFFu) != 0u && rem != 0u;
                    if (lanes_where(names && span_len != 0u) != 0ull) { if (names && span_len != 0u) close_span(px_before); }
                    if (names) span_start = W.ppos;
                    span_len += rem; W.ppos += rem;
                    continue;
                }
                const uint32_t n2 = min(rem, 2u);
                W.put2n(px, n2);
                if (__builtin_expect(lm[u] != 0ull, 0)) {                  // QOI_OP_RUN of three or more (qoi.h:573-575) in some lane
                    rem -= n2;
                    if (rem) {
                        if (rem >= kLongRun) {
                            ++n_long;
                            if (p.desc_all) {
                                // sprites, screenshots with photographs in them: the whole run (the two pixels above taken back) becomes a span for
                                // dec_expand_runs, and the QOI_OP_RUNs that follow it directly lengthen the span - three additions per chunk of 62
                                // pixels.  (First form, a descriptor per chunk with head and tail through the ring: a transparent band of 77 000
                                // pixels was 1239 x (align, ring out, descriptor, tail) in one lane while 63 waited - dec_segments_rec 16.4 ms per
                                // 256 sprite frames against 4.8 for photographs, profiles/r05_s18_dec_kernels.txt.)  A span is written down when
                                // the next one opens or the segment ends; the ring goes out once per span.
                                if (span_len != 0u && W.fpos + n2 == W.ppos && span_start + span_len + n2 == W.ppos && span_px == px) {
                                    W.ppos += rem; span_len += rem + n2; W.fpos = W.ppos; rem = 0u;
                                } else if (n_desc + 1u < kSummaryDescs) {
                                    if (span_len != 0u) { put_desc(n_desc, span_start, span_len, span_px); ++n_desc; }
                                    W.ppos -= n2;
                                    W.finish();
                                    span_start = W.ppos; span_len = rem + n2; span_px = px;
                                    W.ppos += span_len; W.fpos = W.ppos; rem = 0u;
                                } else W.splat(px, rem);
                            } else W.splat(px, rem);
                        }
                        while (rem) { W.put(px); --rem; }
                        if (W.ppos - W.fpos > Writer::kRing - 2u * kDrainEvery) W.drain();
                    }
                }
            }
            ring[2u * d] = S.granule(2u * (blk + kDepth)); ring[2u * d + 1u] = S.granule(2u * (blk + kDepth) + 1u);
        };
        static_assert(kDepth >= 1u && kDepth <= 4u, "ring slots are spelled out");
        for (uint32_t blk0 = 0; blk0 < nblk; blk0 += kDepth) {
            block(std::integral_constant<uint32_t, 0u>{}, blk0);
            if constexpr (kDepth > 1u) block(std::integral_constant<uint32_t, 1u>{}, blk0 + 1u);
            if constexpr (kDepth > 2u) block(std::integral_constant<uint32_t, 2u>{}, blk0 + 2u);
            if constexpr (kDepth > 3u) block(std::integral_constant<uint32_t, 3u>{}, blk0 + 3u);
        }
    };
    if (lanes_where(clip_lane)) run(std::true_type{}); else run(std::false_type{});
    if (FLAT) { if (span_len != 0u) close_span(px); W.fpos = W.ppos; }
    else if (span_len != 0u) { put_desc(n_desc, span_start, span_len, span_px); ++n_desc; }       // (a place was kept for it)
    if (!FLAT && p.tail_fused && lanes_where(n_long != 0u) != 0ull) {           // (header word 6: dec_fill hands it to the host)
        const uint32_t tot = wave_sum(n_long);
        if (lane == 0) atomicAdd(p.pending + 6, tot);
    }
    if (FLAT || p.desc_all) {
        // segments with descriptors queue up for dec_expand_runs: one returning atomic per wavefront that has any
        const bool some = have && n_desc != 0u;
        const u64 m = lanes_where(some);
        if (m != 0ull) {
            uint32_t base = 0u;
            const uint32_t first = (uint32_t)__builtin_ctzll(m);
            if (lane == first) base = atomicAdd(p.run_queue_n, (uint32_t)__builtin_popcountll(m));
            base = read_lane_dyn(base, first);
            if (some) { p.run_queue[base + count_below(m)] = q; p.run_cnt[q] = n_desc; }
        }
    }
    if (have) {
        W.finish();
        if (j + 1u < im.n_active) {
            // exit state must equal what the next segment was started from
            const uint32_t* __restrict__ nxt = ent + 65u;
            bool same = nxt[64] == px;
            for (uint32_t k0 = 0; k0 < 64u; k0 += 16u) {
                uint32_t v[16];
#pragma unroll
                for (uint32_t k = 0; k < 16u; ++k) v[k] = nxt[k0 + k];
#pragma unroll
                for (uint32_t k = 0; k < 16u; ++k) same = same && (v[k] == tab.get(k0 + k));
            }
            if (!same) {
                uint32_t* fx = p.fix + (size_t)(q + 1u) * 65u;
                for (uint32_t k = 0; k < 64u; ++k) fx[k] = tab.get(k);
                fx[64] = px;
                atomicMin(&p.first_bad[img], j + 1u);
            }
        }
    }
    if (have && j + 1u >= im.n_active) p.images[img].final_px = px;     // pixel repeated when the stream ends early (qoi.h:544)
}

// The run descriptors of a round, written out.  A wavefront takes the queued segments wave-th, wave + N-th, ... (N wavefronts in the
// launch; the queue length is only known on the device): 64 descriptors at a time in registers (lane i holds descriptor i, the next
// 64 on their way), every run written by all 64 lanes, 16 bytes (OCH 3: 12 bytes, four pixels) per lane and store instruction - KiB-sized
// pieces of one image row after the other instead of 64 lanes 16 bytes each at 64 places.
// (The first form gave a wavefront the 64 segments of the dec_segments_rec wavefront that made them: 1024 UI frames at 4 KiB segments
// were 2600 wavefronts of 87 000 descriptors each, one after the other: 28.8 ms for 34 GB, profiles/r05_s2_dec_desc.txt.)
constexpr uint32_t kExpandBlocks = 4096;
template <int OCH>
__global__ __launch_bounds__(256) void dec_expand_runs(DecParams p) {
    const uint32_t lane = lane_id(), wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint32_t n_items = *p.run_queue_n;
    const uint32_t n_waves = gridDim.x * 4u;
    for (uint32_t item = blockIdx.x * 4u + wave; item < n_items; item += n_waves) {
        const uint32_t q = __builtin_amdgcn_readfirstlane(p.run_queue[item]);
        const uint32_t img = find_image(p.images, p.n_images, q);
        const uint32_t seg_base = p.images[img].seg_base, desc_base = p.images[img].desc_base;
        const uint32_t n = p.run_cnt[q];
        const uint4* __restrict__ dsc = desc_base != kNoRunDesc ? p.run_desc + (size_t)(desc_base + (q - seg_base)) * p.desc_cap
                                                                 : reinterpret_cast<const uint4*>(p.summary + (size_t)q * 65u);
        uint8_t* __restrict__ out = p.pixels + (size_t)p.images[img].out_index * p.pixel_stride;
        auto get_desc = [&](uint32_t at) {                   // (two 8-byte halves: see dec_segments_rec)
            const uint2* d = reinterpret_cast<const uint2*>(dsc + at);
            const uint2 a = d[0], b = d[1];
            return make_uint4(a.x, a.y, b.x, 0u);
        };
        uint4 cur = lane < n ? get_desc(lane) : make_uint4(0u, 0u, 0u, 0u);
        for (uint32_t d0 = 0; d0 < n; d0 += 64u) {
            const uint4 nxt = d0 + 64u + lane < n ? get_desc(d0 + 64u + lane) : make_uint4(0u, 0u, 0u, 0u);
            const uint32_t m = min(n - d0, 64u);
            for (uint32_t i = 0; i < m; ++i) {
                const uint32_t start = read_lane_dyn(cur.x, i), len = read_lane_dyn(cur.y, i), px = read_lane_dyn(cur.z, i);
                if (OCH == 4) {
                    // pixels up to the first 16-byte boundary and behind the last one as dwords (one store instruction for both ends),
                    // the aligned middle 16 bytes per lane
                    // (from the ADDRESS, not from the pixel index: an image need not begin on a 16-byte boundary)
                    uint32_t* __restrict__ o32 = reinterpret_cast<uint32_t*>(out) + start;
                    const uint32_t head = min(len, (4u - ((uint32_t)(reinterpret_cast<uintptr_t>(o32) >> 2) & 3u)) & 3u), mid = (len - head) >> 2, tail = (len - head) & 3u;
                    if (lane < head + tail) o32[lane < head ? lane : head + (mid << 2) + (lane - head)] = px;
                    uint4* __restrict__ o = reinterpret_cast<uint4*>(o32 + head);
                    const uint4 w4 = make_uint4(px, px, px, px);
                    for (uint32_t k = lane; k < mid; k += 64u) o[k] = w4;
                } else {
                    // 3-byte pixels: bytes up to the first dword boundary and behind the last one singly, the middle as dwords of the
                    // repeating r,g,b pattern (the dword k dwords in starts (head + k) % 3 bytes into a pixel)
                    const uint32_t b0 = start * 3u, nb = len * 3u;
                    const uint32_t head = min(nb, (4u - ((uint32_t)reinterpret_cast<uintptr_t>(out + b0) & 3u)) & 3u), mid = (nb - head) >> 2, tail = (nb - head) & 3u;
                    const uint32_t c0 = px & 0xFFu, c1 = (px >> 8) & 0xFFu, c2 = (px >> 16) & 0xFFu;
                    auto byte_at = [&](uint32_t ph) { return ph == 0u ? c0 : (ph == 1u ? c1 : c2); };
                    uint8_t* __restrict__ o8 = out + b0;
                    if (lane < head + tail) {
                        const uint32_t at = lane < head ? lane : head + (mid << 2) + (lane - head);
                        o8[at] = (uint8_t)byte_at(at % 3u);
                    }
                    uint32_t* __restrict__ o = reinterpret_cast<uint32_t*>(o8 + head);
                    const uint32_t w0 = c0 | (c1 << 8) | (c2 << 16) | (c0 << 24), w1 = c1 | (c2 << 8) | (c0 << 16) | (c1 << 24), w2 = c2 | (c0 << 8) | (c1 << 16) | (c2 << 24);
                    for (uint32_t k = lane; k < mid; k += 64u) { const uint32_t ph = (head + k) % 3u; o[k] = ph == 0u ? w0 : (ph == 1u ? w1 : w2); }
                }
            }
            cur = nxt;
        }
    }
}

// Last resort of the repair loop.  Every round verifies at least one more segment per image, so the loop terminates - but a
// stream built to defeat the speculation (say, QOI_OP_INDEX on slots whose content does not hash there, segment after
// segment) could ask for as many rounds as it has segments, each a relaunch over all of them: quadratic.  After
// kMaxSpecRounds rounds the images that are still open are finished here instead: ONE lane per image walks the remaining
// chunk records in order from the last verified state - a plain qoi.h:540-587 decode, linear in the stream whatever it
// holds (about the speed of one CPU core; encoder-made content never gets here).
template <int OCH>
__global__ __launch_bounds__(64) void dec_sequential(DecParams p) {
    __shared__ uint32_t s_tab[64];
    const uint32_t img = blockIdx.x, lane = lane_id();
    const DecImage im = p.images[img];
    if (im.start_seg >= im.n_active) return;                           // verified already
    const size_t q0 = (size_t)im.seg_base + im.start_seg;
    s_tab[lane] = p.entry[q0 * 65u + lane];                           // the TRUE state at start_seg (dec_prepare_restart)
    uint32_t px = p.entry[q0 * 65u + 64u];
    __builtin_amdgcn_wave_barrier();
    if (lane != 0u) return;
    uint8_t* out = p.pixels + (size_t)im.out_index * p.pixel_stride;
    uint32_t pos = p.px_off[q0], stash = 0u;
    const uint32_t limit = im.npx;
    for (uint32_t j = im.start_seg; j < im.n_active && pos < limit; ++j) {
        const uint32_t q = im.seg_base + j;
        const uint32_t* col = p.recs + ((size_t)(q >> 6) * p.rec_rows * 64u + (q & 63u)) * 4u;      // granule g of this segment at col + g*256
        const uint32_t gw = p.rec_gran[q];
        const uint32_t n0 = p.tr_split ? gw & 0xFFFFu : gw, n = p.tr_split ? n0 + (gw >> 16) : gw;      // (two-lane transcoder: RecSource)
        for (uint32_t g = 0; g < n && pos < limit; ++g) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(col + (size_t)(g < n0 ? g : g - n0 + p.tr_rows_half) * 256u);
            const uint32_t rr[4] = {v.x, v.y, v.z, v.w};
            for (uint32_t k = 0; k < 4u && pos < limit; ++k) {
                const uint32_t rec = rr[k], cls = rec_class(rec);
                if (cls == 2u && rec_pixels(rec) == kRecStash) { stash = rec & 0x00FFFFFFu; continue; }
                const uint32_t t = s_tab[rec & 63u];
                px = cls == 0u ? add_bytes(px, rec & 0x00FFFFFFu) : cls == 1u ? t : cls == 2u ? (px & 0xFF000000u) | (rec & 0x00FFFFFFu) : stash | ((rec & 0xFFu) << 24);
                s_tab[hash_px(px)] = px;                                // qoi.h:577
                uint32_t stop = pos + rec_pixels(rec);
                if (stop > limit) stop = limit;                         // over-long run clipped (Appendix B item 8)
                for (; pos < stop; ++pos) {
                    if (OCH == 4) reinterpret_cast<uint32_t*>(out)[pos] = px;
                    else { uint8_t* d = out + (size_t)pos * 3u; d[0] = (uint8_t)px; d[1] = (uint8_t)(px >> 8); d[2] = (uint8_t)(px >> 16); }
                }
            }
        }
    }
    p.images[img].final_px = px;
    p.images[img].start_seg = im.n_active;
}

// Concrete start state of every image: {0,0,0,255} and a zeroed table (qoi.h:533-537).
// After a round: images whose check failed restart at the first bad segment from the
// TRUE exit state of its predecessor; the others are finished.  Counts pending images.
__device__ __forceinline__ void prepare_restart(const DecParams& p, uint32_t img, uint32_t lane) {
    const DecImage im = p.images[img];
    const uint32_t fb = __hip_atomic_load(&p.first_bad[img], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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

// Pixels the chunks never reach repeat the last pixel (truncated streams, size==22).  One block per (image, slice): the
// image index travels in blockIdx.x (gridDim.y stops at 65535 - a batch may hold more images than that).  The first wavefront
// of an image's first slice also prepares the image's restart (above): one launch less per round.
constexpr uint32_t kFillSlices = 64;
template <int OCH>
__global__ __launch_bounds__(256) void dec_fill(DecParams p) {
    const uint32_t img = blockIdx.x / kFillSlices, slice = blockIdx.x % kFillSlices;
    const DecImage im = p.images[img];
    if (p.tail_fused) {
        // calls of a few images: the launch's first wavefront prepares every image's restart and leaves the round's counters in pinned HOST
        // words (system-scope stores; the host waits for the stream and reads them - no copy back: 4 of a 4K frame's 205 us)
        if (blockIdx.x == 0u && threadIdx.x < 64u) {
            for (uint32_t i = 0; i < p.n_images; ++i) prepare_restart(p, i, threadIdx.x);
            __builtin_amdgcn_wave_barrier();
            const uint32_t pend = __hip_atomic_load(p.pending, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), redo = __hip_atomic_load(p.redo_segs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const uint32_t sf = *p.sync_fails;
            if (threadIdx.x == 0u) {
                uint32_t need_fill = 0u;                                      // some image's chunks end before its pixels do: this launch still has pixels to write
                for (uint32_t i = 0; i < p.n_images; ++i) need_fill |= p.images[i].total_px < p.images[i].npx ? 1u : 0u;
                __hip_atomic_store(&p.host_result[1], redo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&p.host_result[2], sf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&p.host_result[0], pend, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&p.host_result[3], need_fill, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(&p.host_result[5], __hip_atomic_load(p.pending + 6, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // long runs met (dec_segments_rec)
                __hip_atomic_store(&p.host_result[4], p.epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);      // "the words above are this call's"
            }
            // The call is complete (no image to restart, every segment synchronised): the counter header as the context's NEXT call wants to
            // find it - that call then needs no copy in front of its first kernel.  Nothing of this launch looks at these words any more.
            if (pend == 0u && sf == 0u) p.pending[threadIdx.x] = 0u;
            else if (threadIdx.x == 3u || threadIdx.x == 6u) p.pending[threadIdx.x] = 0u;     // run_queue_n (dec_expand_runs ran before this launch), the count of long runs
        }
    } else if (slice == 0u && threadIdx.x < 64u && p.total_segs != 0u) prepare_restart(p, img, threadIdx.x);
    if (blockIdx.x == 0u && threadIdx.x == 64u && !p.tail_fused) *p.run_queue_n = 0u;        // the round's run descriptors are written out (dec_expand_runs ran before this launch)
    if (im.total_px >= im.npx) return;
    const uint32_t px = im.n_active ? im.final_px : kInitPx;
    uint8_t* out = p.pixels + (size_t)im.out_index * p.pixel_stride;
    for (uint32_t i = im.total_px + slice * 256u + threadIdx.x; i < im.npx; i += kFillSlices * 256u) {
        if (OCH == 4) reinterpret_cast<uint32_t*>(out)[i] = px;
        else { uint8_t* d = out + (size_t)i * 3u; d[0] = (uint8_t)px; d[1] = (uint8_t)(px >> 8); d[2] = (uint8_t)(px >> 16); }
    }
}

// ---------------------------------------------------------------------------------
// host-side launchers
// ---------------------------------------------------------------------------------
void launch_decode_parse(const DecParams& p, hipStream_t st, KernelTimer* tm) {
    tm->mark(kT_begin, st);
    if (p.total_segs) {
        if (!p.sync_all) {
            // parse + transcode in one walk (dec_transcode<0>); the five-phase parse only for the segments it flags
            hipLaunchKernelGGL(dec_transcode<0>, dim3((p.total_segs + kTrThreads - 1u) / kTrThreads), dim3(kTrThreads), 0, st, p);
            tm->mark(kT_dec_slot_walk, st);
            hipLaunchKernelGGL(dec_parse_fine, dim3((p.total_segs * p.fine_per_seg + 255u) / 256u), dim3(256), 0, st, p);
        } else hipLaunchKernelGGL(dec_parse, dim3((p.total_segs + 255u) / 256u), dim3(256), 0, st, p);   // segment sizes without the piece parse: every segment
        tm->mark(kT_dec_parse, st);
        hipLaunchKernelGGL(dec_chain_parse_l1, dim3(p.total_grps), dim3(64), 0, st, p);
    }
    hipLaunchKernelGGL(dec_chain_parse_l2, dim3(p.n_images), dim3(64 * kL2Waves), 0, st, p);
    if (p.total_segs) hipLaunchKernelGGL(dec_chain_parse_l3, dim3(p.total_grps), dim3(64), 0, st, p);
    tm->mark(kT_dec_chain_parse, st);
}

// Calls of a few images (DecParams::fused): dec_transcode<0>, then everything P3 needs from ONE kernel (dec_scan_entry)
void launch_decode_fused_front(const DecParams& p, hipStream_t st, KernelTimer* tm) {
    tm->mark(kT_begin, st);
    if (p.tr_scan) {                          // ... and the scan on the same launch
        hipLaunchKernelGGL((dec_transcode<0, false, true, true>), dim3((2u * p.total_segs + kTrThreads - 1u) / kTrThreads), dim3(kTrThreads), 0, st, p);
        tm->mark(kT_dec_slot_walk, st);
        return;
    }
    if (p.tr_split) hipLaunchKernelGGL((dec_transcode<0, kTailsInTranscoder, true>), dim3((2u * p.total_segs + kTrThreads - 1u) / kTrThreads), dim3(kTrThreads), 0, st, p);
    else hipLaunchKernelGGL((dec_transcode<0, kTailsInTranscoder>), dim3((p.total_segs + kTrThreads - 1u) / kTrThreads), dim3(kTrThreads), 0, st, p);
    tm->mark(kT_dec_slot_walk, st);
    hipLaunchKernelGGL(dec_scan_entry, dim3(p.total_segs / kScanSegs), dim3(kScanSegs), 0, st, p);
    tm->mark(kT_dec_chain_slots, st);
}
// ... and where dec_transcode<0> could not synchronise every segment (the host reads sync_fails behind the round): the rest of
// launch_decode_parse on the same records; the round that follows is the three-level one
void launch_decode_parse_rest(const DecParams& p, hipStream_t st, KernelTimer* tm) {
    tm->mark(kT_begin, st);
    hipLaunchKernelGGL(dec_parse_fine, dim3((p.total_segs * p.fine_per_seg + 255u) / 256u), dim3(256), 0, st, p);
    tm->mark(kT_dec_parse, st);
    hipLaunchKernelGGL(dec_chain_parse_l1, dim3(p.total_grps), dim3(64), 0, st, p);
    hipLaunchKernelGGL(dec_chain_parse_l2, dim3(p.n_images), dim3(64 * kL2Waves), 0, st, p);
    hipLaunchKernelGGL(dec_chain_parse_l3, dim3(p.total_grps), dim3(64), 0, st, p);
    tm->mark(kT_dec_chain_parse, st);
}

void launch_decode_round(const DecParams& p_in, int out_channels, bool refine, hipStream_t st, KernelTimer* tm) {
    const DecParams& p = p_in;
    if (!p.total_segs) return;
    const uint32_t b64 = (p.total_segs + 63u) / 64u;
    tm->mark(kT_begin, st);
    uint32_t l2_seq = 0, last_pass = 0;
    // (pass: 1 + the number of the refinement pass whose summaries this chain follows, 0 for the chain behind P3 proper - DecParams::conv)
    auto chain_state = [&](uint32_t pass = 0u) {
        DecParams p = p_in;
        p.conv_pass = (p.conv && pass <= 16u) ? pass : 0u;
        if (p.qtr_summary) hipLaunchKernelGGL(dec_chain_state_l1q, dim3(p.total_grps), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(dec_chain_state_l1, dim3(p.total_grps), dim3(64), 0, st, p);
        if (p.grp_prefix && p.s3_ctr) { }                        // (the per-image level rode on dec_chain_state_l1q's launch)
        else if (p.grp_prefix) hipLaunchKernelGGL(dec_chain_state_l2p, dim3(p.n_images * kL2pWgs), dim3(64 * kL2Waves), 0, st, p);
        else if (p.l2_wgs > 1u) hipLaunchKernelGGL(dec_chain_state_l2m, dim3(p.n_images * p.l2_wgs), dim3(64 * kL2Waves), 0, st, p, p.l2_tag_base + l2_seq++);
        else hipLaunchKernelGGL(dec_chain_state_l2, dim3(p.n_images), dim3(64 * kL2Waves), 0, st, p);
        if (p.qtr_summary) hipLaunchKernelGGL(dec_chain_state_l3q, dim3(p.total_grps), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(dec_chain_state_l3, dim3(p.total_grps), dim3(64), 0, st, p);
        tm->mark(kT_dec_chain_state, st);
    };
    if (refine) {
        // A refinement round repeats "summaries from the hinted entry states, entry states from the summaries" refine_inner times
        // before it decodes: a summary made from the hints of the previous round is stale where the new entry states differ in
        // what the hints are read from (UI frames with several alpha levels crawled two segments per round through such
        // stretches: 13 rounds, each with a P4 over every open segment).  P3 + S3 are a fraction of P4 on such streams.
        const uint32_t inner = p.refine_inner ? p.refine_inner : 1u;
        for (uint32_t it = 0; it < inner; ++it) {
            DecParams pr = p;
            pr.conv_pass = (p.conv && it < 16u) ? it + 1u : 0u;
            hipLaunchKernelGGL(dec_summarize_rec<true>, dim3(b64), dim3(64), 0, st, pr);
            tm->mark(kT_dec_summarize, st);
            if (it + 1u < inner) chain_state(it + 1u);
        }
        last_pass = inner;
    } else {
        if (!p.fused) {                       // (fused: dec_scan_entry has left slot_in / alpha_in)
            hipLaunchKernelGGL(dec_transcode<1>, dim3((p.total_segs + kTrThreads - 1u) / kTrThreads), dim3(kTrThreads), 0, st, p);
            hipLaunchKernelGGL(dec_slot_tails, dim3(b64), dim3(64), 0, st, p);
            tm->mark(kT_dec_slot_walk, st);
            hipLaunchKernelGGL(dec_chain_slots_l1, dim3(p.total_grps), dim3(64), 0, st, p);
            hipLaunchKernelGGL(dec_chain_slots_l2, dim3(p.n_images), dim3(64 * kL2Waves), 0, st, p);
            hipLaunchKernelGGL(dec_chain_slots_l3, dim3(p.total_grps), dim3(64), 0, st, p);
            tm->mark(kT_dec_chain_slots, st);
        }
        hipLaunchKernelGGL(dec_summarize_rec<false>, dim3(b64), dim3(64), 0, st, p);
        tm->mark(kT_dec_summarize, st);
    }
    chain_state(last_pass);
    if (!refine && p.first_inner) {
        // Flat images (dec_image_is_flat): their first round nearly always fails at the second or third segment - P2's guess
        // "an INDEX chunk leaves the alpha as it is" is wrong where alpha levels go through the colour table - and a round's P4
        // rewrites every pixel.  A few refinement passes from the speculated entry states settle most of them before it
        // (host rehearsal, 96 UI frames: 24 verify in the first round without, 89 with two passes, 93 with four).
        DecParams pf = p;
        pf.only_flat = 1u;
        for (uint32_t it = 0; it < p.first_inner; ++it) {
            pf.conv_pass = (p.conv && it < 16u) ? it + 1u : 0u;
            hipLaunchKernelGGL(dec_summarize_rec<true>, dim3(b64), dim3(64), 0, st, pf);
            tm->mark(kT_dec_summarize, st);
            chain_state(it + 1u);
        }
    }
    if (out_channels == 4) hipLaunchKernelGGL((dec_segments_rec<4, false>), dim3(b64), dim3(64), 0, st, p);
    else hipLaunchKernelGGL((dec_segments_rec<3, false>), dim3(b64), dim3(64), 0, st, p);
    if (p.flat_segs) {                   // the call holds flat images: their segments (run descriptors instead of lane-written runs), then the runs
        if (out_channels == 4) hipLaunchKernelGGL((dec_segments_rec<4, true>), dim3(b64), dim3(64), 0, st, p);
        else hipLaunchKernelGGL((dec_segments_rec<3, true>), dim3(b64), dim3(64), 0, st, p);
    }
    tm->mark(kT_dec_segments, st);
    if (p.flat_segs || p.desc_all) {
        const uint32_t eb = (p.total_segs + 3u) / 4u < kExpandBlocks ? (p.total_segs + 3u) / 4u : kExpandBlocks;
        if (out_channels == 4) hipLaunchKernelGGL(dec_expand_runs<4>, dim3(eb), dim3(256), 0, st, p);
        else hipLaunchKernelGGL(dec_expand_runs<3>, dim3(eb), dim3(256), 0, st, p);
        tm->mark(kT_dec_expand, st);
    }
    // (the restart of the images whose check failed is prepared by dec_fill, which every round ends with)
}

void launch_decode_sequential(const DecParams& p, int out_channels, hipStream_t st, KernelTimer* tm) {
    tm->mark(kT_begin, st);
    if (out_channels == 4) hipLaunchKernelGGL(dec_sequential<4>, dim3(p.n_images), dim3(64), 0, st, p);
    else hipLaunchKernelGGL(dec_sequential<3>, dim3(p.n_images), dim3(64), 0, st, p);
    tm->mark(kT_dec_segments, st);
}

void launch_decode_fill(const DecParams& p, int out_channels, hipStream_t st, KernelTimer* tm) {
    const dim3 grid(p.n_images * kFillSlices);
    tm->mark(kT_begin, st);
    if (out_channels == 4) hipLaunchKernelGGL(dec_fill<4>, grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL(dec_fill<3>, grid, dim3(256), 0, st, p);
    tm->mark(kT_dec_fill, st);
}

}  // namespace qoimi
