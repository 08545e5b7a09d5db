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
// the chain at that segment from the true exit state (host loop in qoi_host.hip).
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
// Five parse chains start on bytes base+0..4; a chain's next position depends only on the
// byte it stands on, so chains that meet stay together.  All chains sit within 5 bytes of
// the front (their minimum), which only moves forward: one sequential reader serves them.
struct ParseState { uint32_t p0, p1, p2, p3, p4, c0, c1, c2, c3, c4; };
QOIMI_HD void parse_init(ParseState& s, uint32_t base) {
    s.p0 = base; s.p1 = base + 1; s.p2 = base + 2; s.p3 = base + 3; s.p4 = base + 4;
    s.c0 = s.c1 = s.c2 = s.c3 = s.c4 = 0;
}
QOIMI_HD uint32_t parse_front(const ParseState& s) {
    uint32_t m = s.p0 < s.p1 ? s.p0 : s.p1; m = m < s.p2 ? m : s.p2; m = m < s.p3 ? m : s.p3; m = m < s.p4 ? m : s.p4;
    return m;
}
// b = stream byte at the front position m
QOIMI_HD void parse_step(ParseState& s, uint32_t m, uint32_t b) {
    const uint32_t len = chunk_len(b), npx = chunk_pixels(b);
    if (s.p0 == m) { s.p0 += len; s.c0 += npx; }
    if (s.p1 == m) { s.p1 += len; s.c1 += npx; }
    if (s.p2 == m) { s.p2 += len; s.c2 += npx; }
    if (s.p3 == m) { s.p3 += len; s.c3 += npx; }
    if (s.p4 == m) { s.p4 += len; s.c4 += npx; }
}
QOIMI_HD void parse_finish(const ParseState& s, uint32_t base, uint32_t B, ParseRec& r) {
    const uint32_t nom = base + B;
    const uint32_t e0 = s.p0 > nom ? s.p0 - nom : 0, e1 = s.p1 > nom ? s.p1 - nom : 0, e2 = s.p2 > nom ? s.p2 - nom : 0,
                   e3 = s.p3 > nom ? s.p3 - nom : 0, e4 = s.p4 > nom ? s.p4 - nom : 0;
    r.exit_phase = e0 | (e1 << 3) | (e2 << 6) | (e3 << 9) | (e4 << 12);
    r.pixels[0] = s.c0; r.pixels[1] = s.c1; r.pixels[2] = s.c2; r.pixels[3] = s.c3; r.pixels[4] = s.c4;
}
// in: stream base; [base, seg_end) is this segment clipped to the chunk region; B = nominal size
QOIMI_HD void parse_segment(const uint8_t* in, uint32_t base, uint32_t seg_end, uint32_t B, ParseRec& r) {
    ParseState s; parse_init(s, base);
    for (;;) {
        const uint32_t m = parse_front(s);
        if (m >= seg_end) break;
        parse_step(s, m, in[m]);
    }
    parse_finish(s, base, B, r);
}

// =====================================================================================
// P2 — speculative (slot, alpha) transfer of a segment
//   slot_out  = hc + (h_rel ? slot_in : 0) + (h_alpha ? 11*alpha_in : 0)   (mod 64)
//   alpha_out = a_abs ? ac : alpha_in
// INDEX k is taken to leave slot = k and alpha unchanged (true for encoder-made streams
// whose alpha does not change through the table; anything else is caught by P4's check).
// =====================================================================================
struct SlotRec { uint8_t hc, h_rel, h_alpha, a_abs, ac; };
struct SlotState { uint32_t hc, h_rel, h_alpha, a_abs, ac; };

QOIMI_HD uint32_t lin_hash(uint32_t rgb) {   // 3r+5g+7b of packed bytes (alpha ignored)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_udot4(rgb, 0x00070503u, 0u, false);          // one v_dot4_u32_u8
#else
    return (rgb & 0xFF) * 3u + ((rgb >> 8) & 0xFF) * 5u + ((rgb >> 16) & 0xFF) * 7u;
#endif
}
QOIMI_HD void slot_init(SlotState& s) { s.hc = 0; s.h_rel = 1; s.h_alpha = 0; s.a_abs = 0; s.ac = 0; }
QOIMI_HD void slot_step(SlotState& s, const Chunk& c) {
    const uint32_t lrgb = lin_hash(c.rgba);                    // 3r+5g+7b of the payload
    // RGB: slot = lin(rgb) + 11*alpha (alpha absolute if an RGBA was seen, else the entry alpha)
    const uint32_t hc_rgb = lrgb + (s.a_abs ? 11u * s.ac : 0u);
    const uint32_t hc_rgba = lrgb + 11u * (c.rgba >> 24);
    const bool abs_op = c.is_rgb || c.is_rgba || c.is_index;
    s.hc = (c.is_rgb ? hc_rgb : (c.is_rgba ? hc_rgba : (c.is_index ? c.b1 : s.hc + lin_hash(c.delta)))) & 63u;
    s.h_alpha = c.is_rgb ? (s.a_abs ? 0u : 1u) : ((c.is_rgba || c.is_index) ? 0u : s.h_alpha);
    s.h_rel = abs_op ? 0u : s.h_rel;
    s.ac = c.is_rgba ? (c.rgba >> 24) : s.ac;
    s.a_abs = c.is_rgba ? 1u : s.a_abs;
}
QOIMI_HD void slot_finish(const SlotState& s, SlotRec& r) {
    r.hc = (uint8_t)s.hc; r.h_rel = (uint8_t)s.h_rel; r.h_alpha = (uint8_t)s.h_alpha; r.a_abs = (uint8_t)s.a_abs; r.ac = (uint8_t)s.ac;
}
QOIMI_HD void slot_walk_segment(const uint8_t* in, uint32_t pos, uint32_t seg_end, SlotRec& r) {
    SlotState s; slot_init(s);
    while (pos < seg_end) {
        const Chunk c = crack(load8(in + pos));
        slot_step(s, c);
        pos += c.len;
    }
    slot_finish(s, r);
}

QOIMI_HD void slot_apply(const SlotRec& r, uint32_t& slot, uint32_t& alpha) {
    const uint32_t s = (r.hc + (r.h_rel ? slot : 0u) + (r.h_alpha ? 11u * alpha : 0u)) & 63u;
    const uint32_t a = r.a_abs ? r.ac : alpha;
    slot = s; alpha = a;
}
// r = b after a (both as transfers): closed under composition
QOIMI_HD SlotRec slot_compose(const SlotRec& a, const SlotRec& b) {
    SlotRec r;
    const uint32_t alpha_term = b.h_alpha ? (a.a_abs ? 11u * a.ac : 0u) : 0u;       // b reads the alpha a leaves
    r.hc = (uint8_t)((b.hc + (b.h_rel ? a.hc : 0u) + alpha_term) & 63u);
    r.h_rel = (uint8_t)(b.h_rel && a.h_rel);
    r.h_alpha = (uint8_t)(((b.h_rel && a.h_alpha) || (b.h_alpha && !a.a_abs)) ? 1u : 0u);
    r.a_abs = (uint8_t)(b.a_abs || a.a_abs);
    r.ac = b.a_abs ? b.ac : a.ac;
    return r;
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
    // bit k of the mask to bit 8k (the four shifted copies do not overlap), then 1 -> 0xFF per byte
    const uint32_t ones = ((absmask & 0xFu) * 0x00204081u) & 0x01010101u;
    return (ones << 8) - ones;
}
// concrete value of a symbolic word given the concrete source value
QOIMI_HD uint32_t sym_eval(sym_t s, uint32_t src_value) {
    const uint32_t m = abs_bytemask(sym_abs(s));
    return add_bytes(src_value & ~m, sym_c(s));
}
// symbolic word b (expressed in the exit state of an earlier summary) re-expressed in that
// summary's entry state; a_src = the earlier summary's word that b's source points at
QOIMI_HD sym_t sym_compose(sym_t b, sym_t a_src) {
    const uint32_t mb = abs_bytemask(sym_abs(b));
    const uint32_t c = add_bytes(sym_c(a_src) & ~mb, sym_c(b));
    return sym_make(c, sym_src(a_src), sym_abs(b) | sym_abs(a_src));
}

// After every chunk the decoder stores the pixel at index[hash(pixel)] (qoi.h:577).  A QOI_OP_RUN chunk leaves the
// pixel as it is, so - except as the very first chunk of a stream, when the start pixel is not in the table yet -
// its store rewrites what the previous chunk stored: the summary skips it (the word goes to the spare row
// kSymParkRow).  This matters: a segment that begins inside a run would otherwise store the entry pixel at the
// SPECULATED entry slot, and one wrong guess (the entry pixel taken from a table word that an earlier
// mis-speculation misplaced) used to spoil a table word for all later segments; frames whose alpha changes through
// the colour table needed up to 90 restart rounds for that reason alone, 2-3 with the store skipped.  The
// refinement rounds (entry slot taken from the hinted entry pixel) skip it; round 1 (entry slot from S2) does not need to.
constexpr uint32_t kSymParkRow = 64u;
constexpr uint32_t kLutRunBit = 1u << 15;         // lut_entry: info bit 15
struct SymState { uint32_t pc, ph, slot, alpha, runmask; };   // running pixel: constants, source|absmask<<8; speculated slot/alpha; kLutRunBit unless first segment
// Tab: accessor with get(slot) / set(slot, sym_t).
template <class Tab>
QOIMI_HD void sym_init(SymState& s, uint32_t slot, uint32_t alpha, Tab& tab, bool stream_start) {
    for (uint32_t k = 0; k < 64; ++k) tab.set(k, sym_make(0u, k, 0u));
    s.pc = 0u; s.ph = 64u; s.slot = slot; s.alpha = alpha; s.runmask = stream_start ? 0u : kLutRunBit;
}
template <class Tab>
QOIMI_HD void sym_step(SymState& s, const Chunk& c, Tab& tab) {
    const sym_t t = tab.get(c.b1 & 63u);                         // read unconditionally, used if INDEX
    const uint32_t rgb = c.rgba & 0x00FFFFFFu;
    const uint32_t pc_rel = add_bytes(s.pc, c.delta);            // DIFF/LUMA/RUN (delta 0)
    s.pc = c.is_rgba ? c.rgba : (c.is_rgb ? ((s.pc & 0xFF000000u) | rgb) : (c.is_index ? (uint32_t)t : pc_rel));
    s.ph = c.is_rgba ? (15u << 8) : (c.is_rgb ? (s.ph | (7u << 8)) : (c.is_index ? (uint32_t)(t >> 32) : s.ph));
    const uint32_t s_rgb = lin_hash(rgb) + 11u * s.alpha;
    const uint32_t s_rgba = lin_hash(rgb) + 11u * (c.rgba >> 24);
    s.slot = (c.is_rgb ? s_rgb : (c.is_rgba ? s_rgba : (c.is_index ? c.b1 : s.slot + lin_hash(c.delta)))) & 63u;
    s.alpha = c.is_rgba ? (c.rgba >> 24) : s.alpha;
    tab.set((c.is_run && s.runmask) ? kSymParkRow : s.slot, (sym_t)s.pc | ((sym_t)s.ph << 32));   // index update after every chunk (qoi.h:577)
}
QOIMI_HD sym_t sym_pixel(const SymState& s) { return (sym_t)s.pc | ((sym_t)s.ph << 32); }

// Processes every chunk that starts in [pos, seg_end).  slot/alpha: speculated entry values.
template <class Tab>
QOIMI_HD sym_t summarize_segment(const uint8_t* in, uint32_t pos, uint32_t seg_end,
                                 uint32_t slot, uint32_t alpha, Tab& tab, bool stream_start) {
    SymState s; sym_init(s, slot, alpha, tab, stream_start);
    while (pos < seg_end) {
        const Chunk c = crack(load8(in + pos));
        sym_step(s, c, tab);
        pos += c.len;
    }
    return sym_pixel(s);
}

// =====================================================================================
// P4 — genuine decode of one segment from a concrete entry state (qoi.h:540-587)
// =====================================================================================
// one chunk: new pixel, table updated (qoi.h:547-577).  Tab32: get(slot)/set(slot,uint32_t)
template <class Tab32>
QOIMI_HD uint32_t pixel_step(uint32_t px, const Chunk& c, Tab32& tab) {
    const uint32_t t = tab.get(c.b1 & 63u);                      // read unconditionally, used if INDEX
    const uint32_t rel = add_bytes(px, c.delta);
    px = c.is_rgba ? c.rgba : (c.is_rgb ? ((px & 0xFF000000u) | (c.rgba & 0x00FFFFFFu)) : (c.is_index ? t : rel));
    tab.set(hash_px(px), px);
    return px;
}
QOIMI_HD uint32_t chunk_run(const Chunk& c) { return c.is_run ? (c.b1 & 0x3Fu) + 1u : 1u; }   // qoi.h:573-575

// Writes pixels [px_pos, px_limit) at most.  OCH = output channels (3 or 4).
// Returns the exit pixel; tab holds the exit table.
template <int OCH, class Tab32>
QOIMI_HD uint32_t decode_segment(const uint8_t* in, uint32_t pos, uint32_t seg_end,
                                 uint32_t px, Tab32& tab, uint8_t* out,
                                 uint32_t px_pos, uint32_t px_limit) {
    while (pos < seg_end && px_pos < px_limit) {
        const Chunk c = crack(load8(in + pos));
        px = pixel_step(px, c, tab);
        pos += c.len;
        uint32_t stop = px_pos + chunk_run(c);
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


// =====================================================================================
// Lean per-chunk primitives used by the kernels (the functions above are the readable
// statement of the same steps; tests/host/decode_host.cpp checks both against the oracle).
//
// Everything that depends on the tag byte alone comes from a 256-entry table (LDS on the
// device): a chunk step is then a handful of selects instead of a branch-free re-derivation
// of every field (PMC: the first version spent 92..130 VALU ops per chunk and pass).
//   delta   packed (dr,dg,db,0): DIFF delta; LUMA base (vg-8, vg, vg-8); else 0
//   info    bits 0..2   chunk length (1,2,4,5)                         qoi.h:547-575
//           bits 3..8   pixels produced (1, RUN: 1..62)               qoi.h:573-575
//           bits 9..14  slot shift of the chunk: lin_hash(delta) & 63 (LUMA: of its base)
//           bit  15     RUN (qoi.h:573-575)
//           bit  28     LUMA (second byte adds (b2>>4) to r and (b2&15) to b, qoi.h:566-571)
//           bits 30,31  op class: 0 relative (DIFF/LUMA/RUN), 1 INDEX, 2 RGB, 3 RGBA
// =====================================================================================
QOIMI_HD void lut_entry(uint32_t b, uint32_t& delta, uint32_t& info) {
    const uint32_t top = b >> 6;
    uint32_t len = 1, npx = 1, code = 0, luma = 0, run = 0;
    delta = 0;
    if (b == 0xFEu) { len = 4; code = 2; }
    else if (b == 0xFFu) { len = 5; code = 3; }
    else if (top == 0u) { code = 1; }
    else if (top == 1u) { delta = diff_delta(b); }
    else if (top == 2u) { len = 2; luma = 1; delta = luma_delta(b, 0u); }
    else { npx = (b & 0x3Fu) + 1u; run = 1; }
    info = len | (npx << 3) | ((lin_hash(delta) & 63u) << 9) | (run << 15) | (luma << 28) | (code << 30);
}
QOIMI_HD uint32_t lut_len(uint32_t info) { return info & 7u; }
QOIMI_HD uint32_t lut_pixels(uint32_t info) { return (info >> 3) & 63u; }
QOIMI_HD uint32_t lut_slot_shift(uint32_t info) { return (info >> 9) & 63u; }
QOIMI_HD bool lut_hi(uint32_t info) { return (int32_t)info < 0; }              // RGB or RGBA
QOIMI_HD bool lut_lo(uint32_t info) { return (int32_t)(info << 1) < 0; }       // INDEX or RGBA
// chunk length from the tag byte by arithmetic (keeps the LDS round trip of the table out of the
// byte-cursor's dependency chain)
QOIMI_HD uint32_t len_of(uint32_t b1) {
    const uint32_t base = (0x1211u >> ((b1 >> 6) << 2)) & 0xFu;        // 1,1,2,1 by the two top bits
    return b1 >= 0xFEu ? b1 - 0xFAu : base;                             // 0xFE -> 4, 0xFF -> 5
}
// second byte of a LUMA chunk as packed (b2>>4, 0, b2&15, 0); 0 for every other chunk.  w32 = chunk bytes 0..3
QOIMI_HD uint32_t luma_extra(uint32_t w32, uint32_t info) {
    const uint32_t m = (uint32_t)(((int32_t)(info << 3)) >> 31);        // all ones for LUMA
    return (((w32 >> 12) & 0x0000000Fu) | ((w32 << 8) & 0x000F0000u)) & m;
}

// ---- P1, single chain (after the five entry-phase chains have met) --------------------
// ---- P2: speculative slot/alpha transfer, same function as slot_step ------------------
struct SlotFast { uint32_t hc, ac, fl; };     // fl: bit0 h_rel, bit1 h_alpha, bit2 a_abs
QOIMI_HD void slotf_init(SlotFast& s) { s.hc = 0; s.ac = 0; s.fl = 1u; }
QOIMI_HD void slotf_step(SlotFast& s, uint32_t w32, uint32_t b5, uint32_t info) {
    const uint32_t b1 = w32 & 0xFFu;
    const uint32_t rgb = w32 >> 8;                                       // r,g,b (RGB / RGBA payload)
    const uint32_t lrgb = lin_hash(rgb);
    const uint32_t ex = luma_extra(w32, info);
    const uint32_t rel = s.hc + lut_slot_shift(info) + lin_hash(ex);       // ex = (b2>>4, 0, b2&15): 3*(b2>>4) + 7*(b2&15)
    const bool a_abs = (s.fl & 4u) != 0u;
    const uint32_t hc_rgb = lrgb + (a_abs ? 11u * s.ac : 0u);
    const uint32_t hc_rgba = lrgb + 11u * b5;
    const bool hi = lut_hi(info), lo = lut_lo(info);                     // (hi,lo): 00 rel, 01 INDEX, 10 RGB, 11 RGBA
    const uint32_t a = lo ? b1 : rel, b = lo ? hc_rgba : hc_rgb;
    s.hc = (hi ? b : a) & 63u;
    // flags: rel keeps all; INDEX: h_rel=0,h_alpha=0; RGB: h_rel=0,h_alpha=!a_abs; RGBA: h_rel=0,h_alpha=0,a_abs=1
    const uint32_t f_index = s.fl & 4u;
    const uint32_t f_rgb = (s.fl & 4u) | (a_abs ? 0u : 2u);
    const uint32_t fa = lo ? f_index : s.fl, fb = lo ? 4u : f_rgb;
    s.fl = hi ? fb : fa;
    s.ac = (hi && lo) ? b5 : s.ac;
}
// The same step in the shape the kernels' loops want: the part every chunk needs first, the QOI_OP_RGB / QOI_OP_RGBA
// part only if `any_hi` (on the device: some lane of the wavefront stands on such a chunk - one in a thousand in
// natural images; on the host: this chunk is one).  One body for both cases keeps the loop free of merge copies.
QOIMI_HD void slotf_step_split(SlotFast& s, uint32_t w32, uint32_t b5, uint32_t info, bool any_hi) {
    const bool hi = lut_hi(info), lo = lut_lo(info);
    const uint32_t rel = s.hc + lut_slot_shift(info) + lin_hash(luma_extra(w32, info));
    uint32_t hc = lo ? (w32 & 0xFFu) : rel;                              // INDEX : relative
    uint32_t fl = lo ? (s.fl & 4u) : s.fl;
    if (any_hi) {
        const bool a_abs = (s.fl & 4u) != 0u;
        const uint32_t lrgb = lin_hash(w32 >> 8);
        const uint32_t b = lrgb + (lo ? 11u * b5 : (a_abs ? 11u * s.ac : 0u));    // RGBA : RGB
        const uint32_t fb = lo ? 4u : ((s.fl & 4u) | (a_abs ? 0u : 2u));
        hc = hi ? b : hc;
        fl = hi ? fb : fl;
        s.ac = (hi && lo) ? b5 : s.ac;
    }
    s.hc = hc & 63u;
    s.fl = fl;
}
// the same step for a chunk that is known not to be QOI_OP_RGB / QOI_OP_RGBA (the kernels take this form when no
// lane of the wavefront stands on one - natural images carry one such chunk in a thousand)
QOIMI_HD void slotf_step_norgb(SlotFast& s, uint32_t w32, uint32_t info) {
    const uint32_t rel = s.hc + lut_slot_shift(info) + lin_hash(luma_extra(w32, info));
    const bool lo = lut_lo(info);                                        // INDEX
    s.hc = (lo ? (w32 & 0xFFu) : rel) & 63u;
    s.fl = lo ? (s.fl & 4u) : s.fl;
}
QOIMI_HD void slotf_finish(const SlotFast& s, SlotRec& r) {
    r.hc = (uint8_t)s.hc; r.h_rel = (uint8_t)(s.fl & 1u); r.h_alpha = (uint8_t)((s.fl >> 1) & 1u);
    r.a_abs = (uint8_t)((s.fl >> 2) & 1u); r.ac = (uint8_t)s.ac;
}

// ---- P3: symbolic step, same function as sym_step except for the alpha an INDEX chunk leaves --------
// t = tab.get(b1 & 63), read by the caller (issued early on the device).
// An INDEX chunk takes the alpha of the table entry it names (qoi.h:558-560).  Where that entry's alpha is
// absolute the value is known; where it still refers to the segment's entry state, `hint(src)` supplies a
// guess: the entry alpha of the segment (first round) or the alpha the previous round computed for that
// entry word (refinement rounds).  Only the SPECULATED slot of a later QOI_OP_RGB depends on it.
template <class Tab, class Hint>
QOIMI_HD void symf_step(SymState& s, uint32_t w32, uint32_t b5, uint32_t delta0, uint32_t info, sym_t t, Tab& tab, const Hint& hint) {
    const uint32_t b1 = w32 & 0xFFu;
    const uint32_t ex = luma_extra(w32, info);
    const uint32_t delta = add_bytes(delta0, ex);
    const uint32_t rgba = (w32 >> 8) | (b5 << 24), rgb = rgba & 0x00FFFFFFu;
    const bool hi = lut_hi(info), lo = lut_lo(info);
    const uint32_t pc_rel = add_bytes(s.pc, delta);
    const uint32_t pc_rgb = (s.pc & 0xFF000000u) | rgb;
    const uint32_t pa = lo ? (uint32_t)t : pc_rel, pb = lo ? rgba : pc_rgb;
    const uint32_t ha = lo ? (uint32_t)(t >> 32) : s.ph, hb = lo ? (15u << 8) : (s.ph | (7u << 8));
    s.pc = hi ? pb : pa;
    s.ph = hi ? hb : ha;
    const uint32_t lrgb = lin_hash(rgb);
    const uint32_t s_rel = s.slot + lut_slot_shift(info) + lin_hash(ex);
    const uint32_t sa = lo ? b1 : s_rel, sb = lrgb + 11u * (lo ? b5 : s.alpha);
    s.slot = (hi ? sb : sa) & 63u;
    if (hi && lo) s.alpha = b5;                                           // QOI_OP_RGBA
    else if (lo) s.alpha = (sym_abs(t) & 8u) ? (uint32_t)t >> 24 : hint(sym_src(t));   // QOI_OP_INDEX
    tab.set((info & s.runmask) ? kSymParkRow : s.slot, (sym_t)s.pc | ((sym_t)s.ph << 32));   // index update after every chunk (qoi.h:577)
}

// ---- P4: concrete step, same function as pixel_step --------------------------------------
template <class Tab32>
QOIMI_HD uint32_t pixelf_step(uint32_t px, uint32_t w32, uint32_t b5, uint32_t delta0, uint32_t info, uint32_t t, Tab32& tab) {
    const uint32_t delta = add_bytes(delta0, luma_extra(w32, info));
    const uint32_t rgba = (w32 >> 8) | (b5 << 24);
    const uint32_t rel = add_bytes(px, delta);
    const uint32_t rgbv = (px & 0xFF000000u) | (rgba & 0x00FFFFFFu);
    const bool hi = lut_hi(info), lo = lut_lo(info);
    const uint32_t a = lo ? t : rel, b = lo ? rgba : rgbv;
    px = hi ? b : a;
    tab.set(hash_px(px), px);
    return px;
}

// Plain-pointer reader + table accessors for the host rehearsal and for slow paths.
struct PtrReader {
    const uint8_t* in;
    QOIMI_HD void peek(uint32_t pos, uint32_t& w32, uint32_t& b5) const {
        const unsigned long long w = load8(in + pos);
        w32 = (uint32_t)w; b5 = (uint32_t)(w >> 32) & 0xFFu;
    }
};
struct ChunkLutRef {            // host: plain arrays
    uint32_t delta[256], info[256];
    void build() { for (uint32_t b = 0; b < 256; ++b) lut_entry(b, delta[b], info[b]); }
};

// segment walkers on the lean primitives (host rehearsal; the kernels inline the same loops around their LDS reader)
template <class Lut>
QOIMI_HD void slot_walk_segment_fast(const uint8_t* in, uint32_t pos, uint32_t seg_end, const Lut& lut, SlotRec& r) {
    PtrReader R{in};
    SlotFast s; slotf_init(s);
    while (pos < seg_end) {
        uint32_t w32, b5; R.peek(pos, w32, b5);
        { const uint32_t info = lut.info[w32 & 0xFFu]; slotf_step_split(s, w32, b5, info, lut_hi(info)); }
        pos += len_of(w32 & 0xFFu);
    }
    slotf_finish(s, r);
}
template <class Lut, class Tab, class Hint>
QOIMI_HD sym_t summarize_segment_fast(const uint8_t* in, uint32_t pos, uint32_t seg_end, uint32_t slot, uint32_t alpha,
                                      const Lut& lut, Tab& tab, const Hint& hint, bool stream_start) {
    PtrReader R{in};
    SymState s; sym_init(s, slot, alpha, tab, stream_start);
    while (pos < seg_end) {
        uint32_t w32, b5; R.peek(pos, w32, b5);
        const uint32_t b1 = w32 & 0xFFu;
        symf_step(s, w32, b5, lut.delta[b1], lut.info[b1], tab.get(b1 & 63u), tab, hint);
        pos += len_of(b1);
    }
    return sym_pixel(s);
}
template <int OCH, class Lut, class Tab32>
QOIMI_HD uint32_t decode_segment_fast(const uint8_t* in, uint32_t pos, uint32_t seg_end, uint32_t px, const Lut& lut, Tab32& tab,
                                      uint8_t* out, uint32_t px_pos, uint32_t px_limit) {
    PtrReader R{in};
    while (pos < seg_end && px_pos < px_limit) {
        uint32_t w32, b5; R.peek(pos, w32, b5);
        const uint32_t b1 = w32 & 0xFFu, info = lut.info[b1];
        px = pixelf_step(px, w32, b5, lut.delta[b1], info, tab.get(b1 & 63u), tab);
        pos += len_of(b1);
        uint32_t stop = px_pos + lut_pixels(info);
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
// =====================================================================================
// Chunk RECORDS (round 2).  The table-bound passes P3 / P4 no longer walk the byte stream: one pass
// ("transcode", the old P2 walk) turns every chunk into ONE fixed-width 32-bit record, written per segment
// as a contiguous run of 16-byte granules, and P3 / P4 read records with a wave-uniform stride - no byte
// cursor, no chunk table, no LUMA expansion in the two passes that are bound by their serial chains.
//   bits  0..23  payload
//   bits 24..29  pixels the record produces (0..62; 63 = stash marker, see class 2)
//   bits 30..31  class: 0 relative   payload = byte-wise (dr,dg,db), LUMA's second byte folded in; RUN: 0 (qoi.h:561-575)
//                       1 INDEX      payload byte 0 = slot = payload byte 1                          (qoi.h:558-560)
//                       2 RGB        payload = r,g,b; alpha kept                                    (qoi.h:547-551)
//                                    with pixels == 63: first half of QOI_OP_RGBA - r,g,b are stashed, NOTHING
//                                    else happens (no pixel, the table is left exactly as it is)
//                       3 ALPHA      second half of QOI_OP_RGBA: pixel = stashed r,g,b + payload byte 0 as alpha;
//                                    one pixel, table updated                                       (qoi.h:552-557)
// QOI_OP_RGB and QOI_OP_RGBA leave PAIRS of records on even indices: (class 2, null) and (stash, alpha) - see dec_transcode.
// A record of 0 is the null record (padding of a segment's last granule, pair padding, lanes that are through): relative, delta
// 0, no pixels - it re-stores the running pixel where it already is (index[hash(px)] == px after every chunk).
// The stash half must be a true no-op instead: as a stream's FIRST chunk the start pixel {0,0,0,255} is not in the
// table yet and must not be put there.
// =====================================================================================
constexpr uint32_t kRecStash = 63u;
// Records a segment of B stream bytes can hold, padded to whole granules of 4: one per byte at most (every chunk of one
// byte; a QOI_OP_RGB / QOI_OP_RGBA with its pair padding leaves three at most for four / five bytes) - plus TWO, because such a
// chunk that starts on the segment's last byte still leaves all of its records.
QOIMI_HD uint32_t rec_region_dwords(uint32_t B) { return (B + 2u + 3u) & ~3u; }

QOIMI_HD uint32_t rec_class(uint32_t r) { return r >> 30; }
QOIMI_HD uint32_t rec_pixels(uint32_t r) { return (r >> 24) & 63u; }
QOIMI_HD uint32_t rec_make(uint32_t cls, uint32_t npx, uint32_t payload) { return (cls << 30) | (npx << 24) | (payload & 0x00FFFFFFu); }
// template of the record of a chunk whose tag byte is b (payload completed by the transcoder for LUMA / RGB / RGBA)
QOIMI_HD uint32_t rec_template(uint32_t b) {
    const uint32_t top = b >> 6;
    if (b == 0xFEu) return rec_make(2u, 1u, 0u);
    if (b == 0xFFu) return rec_make(2u, kRecStash, 0u);
    if (top == 0u) return rec_make(1u, 1u, b | (b << 8));        // the slot twice: bits 8..13 address the table as they stand
    if (top == 1u) return rec_make(0u, 1u, diff_delta(b));
    if (top == 2u) return rec_make(0u, 1u, luma_delta(b, 0u));
    return rec_make(0u, (b & 0x3Fu) + 1u, 0u);
}
// records of the chunk at w (chunk bytes 0..7, little endian): 1, or 2 for QOI_OP_RGBA
QOIMI_HD uint32_t rec_of_chunk(unsigned long long w, uint32_t out[2]) {
    const uint32_t b1 = (uint32_t)w & 0xFFu, w32 = (uint32_t)w;
    uint32_t r = rec_template(b1);
    if (b1 == 0xFFu) { out[0] = r | ((w32 >> 8) & 0x00FFFFFFu); out[1] = rec_make(3u, 1u, (uint32_t)(w >> 32) & 0xFFu); return 2u; }
    if (b1 == 0xFEu) { out[0] = r | ((w32 >> 8) & 0x00FFFFFFu); return 1u; }
    if ((b1 >> 6) == 2u) {                                              // LUMA: + (b2>>4, 0, b2&15) byte-wise
        const uint32_t b2 = (w32 >> 8) & 0xFFu;
        const uint32_t ex = (b2 >> 4) | ((b2 & 15u) << 16);
        r = (r & 0xFF000000u) | (add_bytes(r & 0x00FFFFFFu, ex) & 0x00FFFFFFu);
    }
    out[0] = r; return 1u;
}
// records a segment of B stream bytes can leave (whole granules), granule rows and run-descriptor capacity that follow from it
// (round 5 built a 16-bit record format behind -DQOIMI_REC16: the transcoder gained 6 %, P3 and P4 lost more to making the 32-bit
// record again than they gained from half the bytes - EXPERIMENTS.md "16-bit records", profiles/r05_s8_*, r05_s9_*; the code is in
// the history at 06d3f17)
QOIMI_HD uint32_t rec_max_records(uint32_t B) { return rec_region_dwords(B); }
QOIMI_HD uint32_t rec_rows_of(uint32_t B) { return rec_region_dwords(B) / 4u; }
// The speculative slot/alpha transfer of a segment (SlotRec, the old P2 walk) read off its RECORDS, last record first.
// QOI_COLOR_HASH is linear mod 64, so the slot the segment leaves is "what the last chunk that names a slot absolutely
// left + the hash shifts of the relative chunks behind it": scanning backwards, shifts add up until an INDEX, an RGB or
// the alpha half of an RGBA is met - a handful of records in natural images (every fifth chunk is an INDEX), all of them
// only where a segment holds no such chunk.  The forward walk cost the transcoder eleven instructions per chunk.
//   a_abs / ac   whether a QOI_OP_RGBA occurred in the segment and the alpha of the last one (kept by the transcoder: those
//                chunks take its rare path anyway)
struct TailState { uint32_t sum, hc, h_alpha, found, want_stash, alpha; };
QOIMI_HD void tail_init(TailState& t) { t.sum = 0; t.hc = 0; t.h_alpha = 0; t.found = 0; t.want_stash = 0; t.alpha = 0; }
QOIMI_HD void tail_step(TailState& t, uint32_t rec, uint32_t a_abs, uint32_t ac) {
    if (t.found) return;
    const uint32_t cls = rec_class(rec), pay = rec & 0x00FFFFFFu;
    if (t.want_stash) {                                  // the stash half in front of the alpha half just seen: QOI_OP_RGBA names slot and alpha
        t.hc = lin_hash(pay) + 11u * t.alpha + t.sum; t.h_alpha = 0; t.found = 1;
    } else if (cls == 0u) {
        t.sum += lin_hash(pay);
    } else if (cls == 1u) {                              // INDEX k: taken to leave slot k (qoi_decode_core.h "P2")
        t.hc = (rec & 63u) + t.sum; t.h_alpha = 0; t.found = 1;
    } else if (cls == 2u) {                              // RGB (a stash half is only ever met through want_stash): alpha as the segment's RGBAs left it
        t.hc = lin_hash(pay) + t.sum + (a_abs ? 11u * ac : 0u); t.h_alpha = a_abs ? 0u : 1u; t.found = 1;
    } else {
        t.want_stash = 1; t.alpha = rec & 0xFFu;
    }
}
QOIMI_HD SlotRec tail_finish(const TailState& t, uint32_t a_abs, uint32_t ac) {
    SlotRec r;
    r.hc = (uint8_t)((t.found ? t.hc : t.sum) & 63u); r.h_rel = (uint8_t)(t.found ? 0u : 1u); r.h_alpha = (uint8_t)t.h_alpha;
    r.a_abs = (uint8_t)(a_abs ? 1u : 0u); r.ac = (uint8_t)ac;
    return r;
}
QOIMI_HD SlotRec slot_rec_from_records(const uint32_t* recs, uint32_t n_gran, uint32_t a_abs, uint32_t ac) {
    TailState t; tail_init(t);
    for (uint32_t i = 4u * n_gran; i-- > 0u && !t.found;) tail_step(t, recs[i], a_abs, ac);
    return tail_finish(t, a_abs, ac);
}

// transcoder of one segment (host rehearsal; the kernel inlines the same walk around its LDS reader): every chunk
// that starts in [pos, seg_end) -> records, padded with null records to a multiple of 4.  Returns the granule count.
// Also leaves the speculative slot transfer of the segment (the old P2 walk, same function as slot_walk_segment_fast).
template <class Lut>
QOIMI_HD uint32_t transcode_segment(const uint8_t* in, uint32_t pos, uint32_t seg_end, const Lut& lut, uint32_t* recs, SlotRec& sr, long long* mismatch = nullptr) {
    PtrReader R{in};
    SlotFast s; slotf_init(s);
    uint32_t n = 0;
    while (pos < seg_end) {
        const unsigned long long w = load8(in + pos);
        uint32_t two[2];
        const uint32_t k = rec_of_chunk(w, two);
        // QOI_OP_RGB / QOI_OP_RGBA leave a PAIR of records that begins on an even index: (record, null) / (stash half, alpha half),
        // behind a null record where the chunk is met on an odd index (dec_transcode does the same; see there)
        const bool hi = ((uint32_t)w & 0xFEu) == 0xFEu;
        if (hi && (n & 1u)) recs[n++] = 0u;
        recs[n++] = two[0]; if (k == 2u) recs[n++] = two[1]; else if (hi) recs[n++] = 0u;
        uint32_t w32, b5; R.peek(pos, w32, b5);
        { const uint32_t info = lut.info[w32 & 0xFFu]; slotf_step_split(s, w32, b5, info, lut_hi(info)); }
        pos += len_of(w32 & 0xFFu);
    }
    while (n & 3u) recs[n++] = 0u;
    SlotRec fwd; slotf_finish(s, fwd);                          // forward walk (round 1's P2): kept here as the cross-check
    sr = slot_rec_from_records(recs, n >> 2, fwd.a_abs, fwd.ac);   // what the kernels do: the tail of the records
    if (mismatch && (sr.hc != fwd.hc || sr.h_rel != fwd.h_rel || sr.h_alpha != fwd.h_alpha || sr.a_abs != fwd.a_abs || (fwd.a_abs && sr.ac != fwd.ac))) ++*mismatch;
    return n >> 2;
}
// P3 on records: same function of the chunks as symf_step.  stash: r,g,b of a pending QOI_OP_RGBA.
template <class Tab, class Hint>
QOIMI_HD void symr_step(SymState& s, uint32_t& stash, uint32_t rec, sym_t t, Tab& tab, const Hint& hint) {
    const uint32_t cls = rec_class(rec), idx = rec & 63u;
    if (cls == 2u && rec_pixels(rec) == kRecStash) { stash = rec & 0x00FFFFFFu; tab.set(idx, t); return; }   // true no-op
    const bool is_run = (rec & 0xC0FFFFFFu) == 0u;                        // RUN, zero-delta chunk or null record: the pixel stays
    if (cls == 0u) {
        s.pc = add_bytes(s.pc, rec & 0x00FFFFFFu);
        s.slot = (s.slot + lin_hash(rec & 0x00FFFFFFu)) & 63u;
    } else if (cls == 1u) {
        s.pc = (uint32_t)t; s.ph = (uint32_t)(t >> 32);
        s.slot = idx;
        s.alpha = (sym_abs(t) & 8u) ? (uint32_t)t >> 24 : hint(sym_src(t));
    } else if (cls == 2u) {
        s.pc = (s.pc & 0xFF000000u) | (rec & 0x00FFFFFFu); s.ph |= 7u << 8;
        s.slot = (lin_hash(rec & 0x00FFFFFFu) + 11u * s.alpha) & 63u;
    } else {
        const uint32_t a = rec & 0xFFu;
        s.pc = stash | (a << 24); s.ph = 15u << 8;
        s.slot = (lin_hash(stash) + 11u * a) & 63u; s.alpha = a;
    }
    if (is_run && s.runmask) tab.set(idx, t);                             // store skipped: the word read is written back
    else tab.set(s.slot, (sym_t)s.pc | ((sym_t)s.ph << 32));              // index update after every chunk (qoi.h:577)
}
template <class Tab, class Hint>
QOIMI_HD sym_t summarize_records(const uint32_t* recs, uint32_t n_gran, uint32_t slot, uint32_t alpha, Tab& tab, const Hint& hint, bool stream_start) {
    SymState s; sym_init(s, slot, alpha, tab, stream_start);
    uint32_t stash = 0;
    for (uint32_t i = 0; i < 4u * n_gran; ++i) symr_step(s, stash, recs[i], tab.get(recs[i] & 63u), tab, hint);
    return sym_pixel(s);
}
// P3 on records in the PLAIN form the kernel uses while no QOI_OP_RGBA has been met: a symbolic value is one dword, the code
// (source 0..64, + 65 if r,g,b are absolute) in the top byte and the r,g,b constants below; the alpha is the source's.
// Same function as symr_step on such records; returns false at the first record that needs the general form.
constexpr uint32_t kPlainRgb = 65u;
struct PlainState { uint32_t ppc, slot; };
template <class Tab32, class Hint>
QOIMI_HD bool symp_step(PlainState& s, uint32_t rec, uint32_t alpha_in0, bool refine, bool skip_runs, Tab32& tab, const Hint& hint) {
    const uint32_t cls = rec_class(rec), idx = rec & 63u;
    if (cls == 3u || (cls == 2u && rec_pixels(rec) == kRecStash)) return false;
    const uint32_t t = tab.get(idx);
    uint32_t npc, nslot;
    if (cls == 0u) { npc = (s.ppc & 0xFF000000u) | (add_bytes(s.ppc, rec) & 0x00FFFFFFu); nslot = s.slot + lin_hash(rec & 0x00FFFFFFu); }
    else if (cls == 1u) { npc = t; nslot = idx; }
    else {
        const uint32_t code = s.ppc >> 24, src = code < kPlainRgb ? code : code - kPlainRgb;
        const uint32_t a_src = refine ? hint(src) : alpha_in0;
        npc = ((src + kPlainRgb) << 24) | (rec & 0x00FFFFFFu);
        nslot = lin_hash(rec & 0x00FFFFFFu) + 11u * a_src;
    }
    s.ppc = npc; s.slot = nslot & 63u;
    const bool keep = skip_runs && (rec & 0xC0FFFFFFu) == 0u;
    tab.set(keep ? idx : s.slot, keep ? t : s.ppc);
    return true;
}
// summarize_records with the plain form for as long as it lasts (host rehearsal of dec_summarize_rec's two forms)
template <class Tab, class Hint>
QOIMI_HD sym_t summarize_records_plain(const uint32_t* recs, uint32_t n_gran, uint32_t slot, uint32_t alpha, Tab& tab, const Hint& hint, bool stream_start, bool refine) {
    struct T32 { uint32_t v[64]; uint32_t get(uint32_t k) const { return v[k]; } void set(uint32_t k, uint32_t x) { v[k] = x; } } pt;
    for (uint32_t k = 0; k < 64u; ++k) pt.v[k] = k << 24;
    PlainState ps; ps.ppc = 64u << 24; ps.slot = slot;
    const bool skip_runs = !stream_start;                       // as SymState::runmask: the refinement rounds of every segment but a stream's first
    uint32_t i = 0;
    for (; i < 4u * n_gran; ++i) if (!symp_step(ps, recs[i], alpha, refine, skip_runs, pt, hint)) break;
    auto expand = [](uint32_t w) { const uint32_t code = w >> 24; return sym_make(w & 0x00FFFFFFu, code < kPlainRgb ? code : code - kPlainRgb, code < kPlainRgb ? 0u : 7u); };
    if (i == 4u * n_gran) {
        for (uint32_t k = 0; k < 64u; ++k) tab.set(k, expand(pt.v[k]));
        return expand(ps.ppc);
    }
    // general form from here on
    SymState st;
    for (uint32_t k = 0; k < 64u; ++k) tab.set(k, expand(pt.v[k]));
    const sym_t px = expand(ps.ppc);
    st.pc = (uint32_t)px; st.ph = (uint32_t)(px >> 32); st.slot = ps.slot;
    { const uint32_t code = ps.ppc >> 24, src = code < kPlainRgb ? code : code - kPlainRgb; st.alpha = refine ? hint(src) : alpha; }
    st.runmask = stream_start ? 0u : kLutRunBit;
    uint32_t stash = 0;
    for (; i < 4u * n_gran; ++i) symr_step(st, stash, recs[i], tab.get(recs[i] & 63u), tab, hint);
    return sym_pixel(st);
}

// P4 on records: same function of the chunks as pixelf_step / decode_segment_fast
template <int OCH, class Tab32>
QOIMI_HD uint32_t decode_records(const uint32_t* recs, uint32_t n_gran, uint32_t px, Tab32& tab, uint8_t* out, uint32_t px_pos, uint32_t px_limit) {
    uint32_t stash = 0;
    for (uint32_t i = 0; i < 4u * n_gran; ++i) {
        const uint32_t rec = recs[i], cls = rec_class(rec), idx = rec & 63u;
        const uint32_t t = tab.get(idx);
        if (px_pos >= px_limit) break;                                  // the decoder has stopped (qoi.h:540)
        if (cls == 2u && rec_pixels(rec) == kRecStash) { stash = rec & 0x00FFFFFFu; continue; }
        px = cls == 0u ? (add_bytes(px, rec & 0x00FFFFFFu) & 0x00FFFFFFu) | (px & 0xFF000000u)
           : cls == 1u ? t
           : cls == 2u ? (px & 0xFF000000u) | (rec & 0x00FFFFFFu)
                       : stash | ((rec & 0xFFu) << 24);
        tab.set(hash_px(px), px);
        uint32_t stop = px_pos + rec_pixels(rec);
        if (stop > px_limit) stop = px_limit;                            // over-long run clipped (Appendix B item 8)
        for (; px_pos < stop; ++px_pos) {
            if (OCH == 4) reinterpret_cast<uint32_t*>(out)[px_pos] = px;
            else { uint8_t* d = out + (size_t)px_pos * 3u; d[0] = (uint8_t)px; d[1] = (uint8_t)(px >> 8); d[2] = (uint8_t)(px >> 16); }
        }
    }
    return px;
}

// P1 with the single-chain fast path: five chains until they have met, then one cursor
template <class Lut>
QOIMI_HD void parse_segment_fast(const uint8_t* in, uint32_t base, uint32_t seg_end, uint32_t B, const Lut& lut, ParseRec& r) {
    ParseState s; parse_init(s, base);
    uint32_t m = base;
    while (m < seg_end && !(s.p0 == s.p1 && s.p1 == s.p2 && s.p2 == s.p3 && s.p3 == s.p4)) {
        parse_step(s, m, in[m]);
        m = parse_front(s);
    }
    uint32_t add = 0;
    while (m < seg_end) {                                   // all five cursors stand on m
        const uint32_t b1 = in[m];
        add += lut_pixels(lut.info[b1]);
        m += len_of(b1);
    }
    if (s.p0 == s.p1 && s.p1 == s.p2 && s.p2 == s.p3 && s.p3 == s.p4) {
        s.p0 = s.p1 = s.p2 = s.p3 = s.p4 = m;
        s.c0 += add; s.c1 += add; s.c2 += add; s.c3 += add; s.c4 += add;
    }
    parse_finish(s, base, B, r);
}

}  // namespace qoimi
