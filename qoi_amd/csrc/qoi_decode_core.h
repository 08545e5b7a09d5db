// qoi_decode_core.h — per-segment decode primitives (one GPU lane = one stream segment).
//
// The reference decoder (qoi.h:488-590) is a sequential state machine over
// (cursor p, run, px, index[64]).  The GPU decoder cuts the chunk region of every
// stream into fixed-size byte SEGMENTS and gives each segment to one lane.  What a
// lane needs to know at its segment start is resolved by three short passes:
//
//   P1 parse      where the first chunk of the segment starts (0..4 bytes in) and how
//                 many pixels precede it.  Chunk length is a function of the first byte
//                 only (qoi.h:547-575), so a segment is summarised for all five possible
//                 entry phases and the summaries are chained per image (exact).
//   P2 slot walk  SPECULATES the hash slot of the running pixel at segment entry
//                 (QOI_COLOR_HASH is linear mod 64, so DIFF/LUMA move it by a constant;
//                 INDEX k sets it to k, which holds for every encoder-produced stream).
//   P3 summary    symbolic execution of the segment: every value is "entry value +
//                 per-channel constant" or an absolute constant; gives the segment's
//                 effect on (px, index[64]) as 65 symbolic words, applied in sequence to
//                 get the concrete entry state of every segment.
//   P4 decode     a GENUINE qoi.h:540-587 decode of the segment from that entry state,
//                 writing pixels, then CHECKS its exit state against the next segment's
//                 entry state.
//
// Exactness never rests on the speculation: if every check passes, induction over the
// segments shows the output equals the sequential decoder's; a failed check restarts
// the chain at that segment from the true exit state (host loop in qoi_host.cpp).
//
// The functions are plain sequential code, compiled for the device by hipcc and — by
// tests/host/decode_host.cpp only — for the host, so their logic is unit-tested on CPU
// against the oracle before it ever runs on a GPU.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define QOIMI_HD __host__ __device__ __forceinline__
#else
#define QOIMI_HD inline
#endif

namespace qoimi {

// ---- chunk grammar (qoi.h:547-575): length and pixel yield from the first byte ------
QOIMI_HD uint32_t chunk_len(uint32_t b) {
    return b == 0xFEu ? 4u : b == 0xFFu ? 5u : ((b & 0xC0u) == 0x80u ? 2u : 1u);
}
QOIMI_HD uint32_t chunk_pixels(uint32_t b) {
    return ((b & 0xC0u) == 0xC0u && b < 0xFEu) ? (b & 0x3Fu) + 1u : 1u;   // RUN: (b1&0x3f)+1, qoi.h:573-575
}
QOIMI_HD uint32_t hash_px(uint32_t px) {                                  // qoi.h:322, & 63
    return ((px & 0xFF) * 3u + ((px >> 8) & 0xFF) * 5u + ((px >> 16) & 0xFF) * 7u + (px >> 24) * 11u) & 63u;
}
// per-byte add mod 256 on packed r,g,b,a (qoi.h:562-571 wrap-around)
QOIMI_HD uint32_t add_bytes(uint32_t a, uint32_t b) {
    return ((a & 0x7F7F7F7Fu) + (b & 0x7F7F7F7Fu)) ^ ((a ^ b) & 0x80808080u);
}
// packed (dr,dg,db,0) of a DIFF / LUMA chunk
QOIMI_HD uint32_t diff_delta(uint32_t b1) {
    const uint32_t dr = (((b1 >> 4) & 3u) - 2u) & 0xFFu, dg = (((b1 >> 2) & 3u) - 2u) & 0xFFu, db = ((b1 & 3u) - 2u) & 0xFFu;
    return dr | (dg << 8) | (db << 16);
}
QOIMI_HD uint32_t luma_delta(uint32_t b1, uint32_t b2) {
    const uint32_t vg = (b1 & 0x3Fu) - 32u;
    const uint32_t dr = (vg - 8u + ((b2 >> 4) & 0x0Fu)) & 0xFFu, dg = vg & 0xFFu, db = (vg - 8u + (b2 & 0x0Fu)) & 0xFFu;
    return dr | (dg << 8) | (db << 16);
}

// One unaligned 8-byte load covers the longest chunk (5 bytes).  Streams handed to the
// decoder must be readable 8 bytes past their last chunk byte (true for every stream that
// carries its 8-byte end marker, qoi.h:103, plus the padding the C-ABI documents).
QOIMI_HD unsigned long long load8(const uint8_t* p) {
    unsigned long long w;
    __builtin_memcpy(&w, p, 8);
    return w;
}

// Fields of the chunk whose first byte is the low byte of w - all cases computed, the
// callers select.  No data-dependent branches: a wavefront's lanes sit on different ops.
struct Chunk {
    uint32_t b1;          // tag byte
    uint32_t rgba;        // bytes 1..4 as r,g,b,a
    uint32_t delta;       // packed (dr,dg,db,0) if DIFF/LUMA, else 0
    uint32_t len;         // chunk length in bytes
    bool is_rgb, is_rgba, is_index, is_rel, is_run;   // is_rel: DIFF or LUMA
};
QOIMI_HD Chunk crack(unsigned long long w) {
    Chunk c;
    c.b1 = (uint32_t)w & 0xFFu;
    c.rgba = (uint32_t)(w >> 8);
    const uint32_t b2 = (uint32_t)(w >> 8) & 0xFFu;
    const uint32_t top = c.b1 >> 6;
    c.is_rgb = c.b1 == 0xFEu;
    c.is_rgba = c.b1 == 0xFFu;
    c.is_index = top == 0u;
    const bool is_diff = top == 1u, is_luma = top == 2u;
    c.is_rel = is_diff || is_luma;
    c.is_run = top == 3u && c.b1 < 0xFEu;
    const uint32_t dd = diff_delta(c.b1), dl = luma_delta(c.b1, b2);
    c.delta = is_diff ? dd : (is_luma ? dl : 0u);
    c.len = c.is_rgb ? 4u : (c.is_rgba ? 5u : (is_luma ? 2u : 1u));
    return c;
}

// =====================================================================================
// P1 — parse summary of one segment for the five entry phases
// =====================================================================================
struct ParseRec {
    uint32_t exit_phase;   // 3 bits per entry phase: phase handed to the next segment
    uint32_t pixels[5];    // pixels produced by the chunks that START in this segment
};

// in: stream base; [base, seg_end) is this segment clipped to the chunk region; B = nominal size
QOIMI_HD void parse_segment(const uint8_t* in, uint32_t base, uint32_t seg_end, uint32_t B, ParseRec& r) {
    uint32_t p0 = base, p1 = base + 1, p2 = base + 2, p3 = base + 3, p4 = base + 4;
    uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0;
    for (;;) {
        uint32_t m = p0 < p1 ? p0 : p1; m = m < p2 ? m : p2; m = m < p3 ? m : p3; m = m < p4 ? m : p4;
        if (m >= seg_end) break;
        const uint32_t b = in[m];
        const uint32_t len = chunk_len(b), npx = chunk_pixels(b);
        // chains standing on the same byte advance together (they have merged)
        if (p0 == m) { p0 += len; c0 += npx; }
        if (p1 == m) { p1 += len; c1 += npx; }
        if (p2 == m) { p2 += len; c2 += npx; }
        if (p3 == m) { p3 += len; c3 += npx; }
        if (p4 == m) { p4 += len; c4 += npx; }
    }
    const uint32_t nom = base + B;
    const uint32_t e0 = p0 > nom ? p0 - nom : 0, e1 = p1 > nom ? p1 - nom : 0, e2 = p2 > nom ? p2 - nom : 0,
                   e3 = p3 > nom ? p3 - nom : 0, e4 = p4 > nom ? p4 - nom : 0;
    r.exit_phase = e0 | (e1 << 3) | (e2 << 6) | (e3 << 9) | (e4 << 12);
    r.pixels[0] = c0; r.pixels[1] = c1; r.pixels[2] = c2; r.pixels[3] = c3; r.pixels[4] = c4;
}

// =====================================================================================
// P2 — speculative (slot, alpha) transfer of a segment
//   slot_out  = hc + (h_rel ? slot_in : 0) + (h_alpha ? 11*alpha_in : 0)   (mod 64)
//   alpha_out = a_abs ? ac : alpha_in
// INDEX k is taken to leave slot = k and alpha unchanged (true for encoder-made streams
// whose alpha does not change through the table; anything else is caught by P4's check).
// =====================================================================================
struct SlotRec { uint8_t hc, h_rel, h_alpha, a_abs, ac; };

QOIMI_HD uint32_t lin_hash(uint32_t rgb) {   // 3r+5g+7b of packed bytes (alpha ignored)
    return (rgb & 0xFF) * 3u + ((rgb >> 8) & 0xFF) * 5u + ((rgb >> 16) & 0xFF) * 7u;
}

QOIMI_HD void slot_walk_segment(const uint8_t* in, uint32_t pos, uint32_t seg_end, SlotRec& r) {
    uint32_t hc = 0, h_rel = 1, h_alpha = 0, a_abs = 0, ac = 0;
    while (pos < seg_end) {
        const Chunk c = crack(load8(in + pos));
        const uint32_t lrgb = lin_hash(c.rgba);                    // 3r+5g+7b of the payload
        // RGB: slot = lin(rgb) + 11*alpha (alpha absolute if an RGBA was seen, else the entry alpha)
        const uint32_t hc_rgb = lrgb + (a_abs ? 11u * ac : 0u);
        const uint32_t hc_rgba = lrgb + 11u * (c.rgba >> 24);
        const bool abs_op = c.is_rgb || c.is_rgba || c.is_index;
        hc = c.is_rgb ? hc_rgb : (c.is_rgba ? hc_rgba : (c.is_index ? c.b1 : hc + lin_hash(c.delta)));
        h_alpha = c.is_rgb ? (a_abs ? 0u : 1u) : ((c.is_rgba || c.is_index) ? 0u : h_alpha);
        h_rel = abs_op ? 0u : h_rel;
        ac = c.is_rgba ? (c.rgba >> 24) : ac;
        a_abs = c.is_rgba ? 1u : a_abs;
        hc &= 63u;
        pos += c.len;
    }
    r.hc = (uint8_t)hc; r.h_rel = (uint8_t)h_rel; r.h_alpha = (uint8_t)h_alpha; r.a_abs = (uint8_t)a_abs; r.ac = (uint8_t)ac;
}

QOIMI_HD void slot_apply(const SlotRec& r, uint32_t& slot, uint32_t& alpha) {
    const uint32_t s = (r.hc + (r.h_rel ? slot : 0u) + (r.h_alpha ? 11u * alpha : 0u)) & 63u;
    const uint32_t a = r.a_abs ? r.ac : alpha;
    slot = s; alpha = a;
}

// =====================================================================================
// P3 — symbolic summary.  A symbolic value is 64 bits:
//   bits  0..31  per-channel constant c (r,g,b,a)
//   bits 32..38  source: 0..63 = entry table slot, 64 = entry pixel
//   bits 40..43  absolute mask: channel = c (bit set) or = source.channel + c (bit clear)
// =====================================================================================
typedef unsigned long long sym_t;
QOIMI_HD sym_t sym_make(uint32_t c, uint32_t src, uint32_t absmask) {
    return (sym_t)c | ((sym_t)src << 32) | ((sym_t)absmask << 40);
}
QOIMI_HD uint32_t sym_c(sym_t s) { return (uint32_t)s; }
QOIMI_HD uint32_t sym_src(sym_t s) { return (uint32_t)(s >> 32) & 0x7Fu; }
QOIMI_HD uint32_t sym_abs(sym_t s) { return (uint32_t)(s >> 40) & 0xFu; }
QOIMI_HD uint32_t abs_bytemask(uint32_t absmask) {     // 4-bit channel mask -> 0xFF per set channel
    return ((absmask & 1u) ? 0x000000FFu : 0u) | ((absmask & 2u) ? 0x0000FF00u : 0u) |
           ((absmask & 4u) ? 0x00FF0000u : 0u) | ((absmask & 8u) ? 0xFF000000u : 0u);
}
// concrete value of a symbolic word given the concrete source value
QOIMI_HD uint32_t sym_eval(sym_t s, uint32_t src_value) {
    const uint32_t m = abs_bytemask(sym_abs(s));
    return add_bytes(src_value & ~m, sym_c(s));
}

// Tab: accessor with get(slot) / set(slot, sym_t).  slot/alpha: speculated entry values.
// Processes every chunk that starts in [pos, seg_end).
template <class Tab>
QOIMI_HD sym_t summarize_segment(const uint8_t* in, uint32_t pos, uint32_t seg_end,
                                 uint32_t slot, uint32_t alpha, Tab& tab) {
    for (uint32_t k = 0; k < 64; ++k) tab.set(k, sym_make(0u, k, 0u));
    uint32_t pc = 0u;               // per-channel constants of the running pixel
    uint32_t ph = 64u;              // its source (bits 0..6) and absolute mask (bits 8..11)
    while (pos < seg_end) {
        const Chunk c = crack(load8(in + pos));
        const sym_t t = tab.get(c.b1 & 63u);                         // read unconditionally, used if INDEX
        const uint32_t rgb = c.rgba & 0x00FFFFFFu;
        // constants
        const uint32_t pc_rel = add_bytes(pc, c.delta);              // DIFF/LUMA/RUN (delta 0)
        pc = c.is_rgba ? c.rgba : (c.is_rgb ? ((pc & 0xFF000000u) | rgb) : (c.is_index ? (uint32_t)t : pc_rel));
        // source / absolute mask
        ph = c.is_rgba ? (15u << 8) : (c.is_rgb ? (ph | (7u << 8)) : (c.is_index ? (uint32_t)(t >> 32) : ph));
        // speculated slot / alpha of the new pixel
        const uint32_t s_rgb = lin_hash(rgb) + 11u * alpha;
        const uint32_t s_rgba = lin_hash(rgb) + 11u * (c.rgba >> 24);
        slot = c.is_rgb ? s_rgb : (c.is_rgba ? s_rgba : (c.is_index ? c.b1 : slot + lin_hash(c.delta)));
        slot &= 63u;
        alpha = c.is_rgba ? (c.rgba >> 24) : alpha;
        tab.set(slot, (sym_t)pc | ((sym_t)ph << 32));                // index update after every chunk (qoi.h:577)
        pos += c.len;
    }
    return (sym_t)pc | ((sym_t)ph << 32);
}

// =====================================================================================
// P4 — genuine decode of one segment from a concrete entry state (qoi.h:540-587)
// =====================================================================================
// Tab32: get(slot)/set(slot,uint32_t).  Writes pixels [px_pos, px_limit) at most.
// OCH = output channels (3 or 4).  Returns the exit pixel; tab holds the exit table.
template <int OCH, class Tab32>
QOIMI_HD uint32_t decode_segment(const uint8_t* in, uint32_t pos, uint32_t seg_end,
                                 uint32_t px, Tab32& tab, uint8_t* out,
                                 uint32_t px_pos, uint32_t px_limit) {
    while (pos < seg_end && px_pos < px_limit) {
        const Chunk c = crack(load8(in + pos));
        const uint32_t t = tab.get(c.b1 & 63u);                      // read unconditionally, used if INDEX
        const uint32_t rel = add_bytes(px, c.delta);
        px = c.is_rgba ? c.rgba : (c.is_rgb ? ((px & 0xFF000000u) | (c.rgba & 0x00FFFFFFu)) : (c.is_index ? t : rel));
        const uint32_t n = c.is_run ? (c.b1 & 0x3Fu) + 1u : 1u;
        pos += c.len;
        tab.set(hash_px(px), px);
        uint32_t stop = px_pos + n;
        if (stop > px_limit) stop = px_limit;            // over-long run clipped (Appendix B item 8)
        for (; px_pos < stop; ++px_pos) {
            if (OCH == 4) {
                reinterpret_cast<uint32_t*>(out)[px_pos] = px;
            } else {
                uint8_t* d = out + (size_t)px_pos * 3u;
                d[0] = (uint8_t)px; d[1] = (uint8_t)(px >> 8); d[2] = (uint8_t)(px >> 16);
            }
        }
    }
    return px;
}

}  // namespace qoimi
