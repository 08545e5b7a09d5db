// decode_host.cpp — HOST rehearsal of the GPU decoder's segment pipeline (TEST ONLY).
//
// Compiles the very same per-segment primitives the kernels use
// (qoi_amd/csrc/qoi_decode_core.h) with g++ and drives them with plain loops in the
// order qoi_decode.hip launches its kernels: P1 parse -> S1 chain -> [P2 slot walk ->
// S2 chain -> P3 summary -> S3 state chain -> P4 decode + exit-state check -> restart]
// until every check passes -> fill.  tests/test_decode_scheme.py runs it over the golden
// decode cases at several segment sizes, so the scheme's exactness (incl. the repair
// loop on streams that defeat the speculation) is established on CPU.  Not part of the
// product library and never a fallback for it.
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../qoi_amd/csrc/qoi_decode_core.h"

using namespace qoimi;

namespace {
struct SymTab { sym_t v[65]; sym_t get(uint32_t k) const { return v[k]; } void set(uint32_t k, sym_t x) { v[k] = x; } };
struct Tab32 { uint32_t v[64]; uint32_t get(uint32_t k) const { return v[k]; } void set(uint32_t k, uint32_t x) { v[k] = x; } };
}

// returns 0; stats[0] = rounds, stats[1] = segments re-decoded, stats[2] = segments
// GRP: segments per group of the two-level chains (the kernels use 64)
// mode 0: readable primitives, 1: lean LUT-driven primitives, 2: chunk records (transcode once, P3 / P4 read records)
static int pipeline(const uint8_t* in, int size, uint32_t npx, int och, uint32_t B, uint32_t GRP,
                    uint8_t* out, long long* stats, int mode);
extern "C" int host_decode_pipeline_g(const uint8_t* in, int size, uint32_t npx, int och, uint32_t B, uint32_t GRP,
                                      uint8_t* out, long long* stats) {
    return pipeline(in, size, npx, och, B, GRP, out, stats, 0);
}
// same pipeline on the lean LUT-driven primitives the kernels use
extern "C" int host_decode_pipeline_fast(const uint8_t* in, int size, uint32_t npx, int och, uint32_t B, uint32_t GRP,
                                         uint8_t* out, long long* stats) {
    return pipeline(in, size, npx, och, B, GRP, out, stats, 1);
}
// the pipeline the round-2 kernels run: one transcode pass leaves fixed-width chunk records, P3 / P4 read those
extern "C" int host_decode_pipeline_rec(const uint8_t* in, int size, uint32_t npx, int och, uint32_t B, uint32_t GRP,
                                        uint8_t* out, long long* stats) {
    return pipeline(in, size, npx, och, B, GRP, out, stats, 2);
}
// rec_of_chunk against crack() for every tag byte and a sweep of payloads; returns mismatches
extern "C" int host_check_records(void) {
    int bad = 0;
    for (uint32_t b = 0; b < 256; ++b)
        for (uint32_t pay = 0; pay < 4096; pay += 37) {
            const unsigned long long w = (unsigned long long)b | ((unsigned long long)(pay * 0x9E3779B1u) << 8) | ((unsigned long long)(pay & 0xFF) << 40);
            const Chunk c = crack(w);
            uint32_t r[2];
            const uint32_t k = rec_of_chunk(w, r);
            if (k != (c.is_rgba ? 2u : 1u)) ++bad;
            if (c.is_rel && (rec_class(r[0]) != 0u || (r[0] & 0xFFFFFFu) != c.delta || rec_pixels(r[0]) != 1u)) ++bad;
            if (c.is_run && (rec_class(r[0]) != 0u || (r[0] & 0xFFFFFFu) != 0u || rec_pixels(r[0]) != chunk_run(c))) ++bad;
            if (c.is_index && (rec_class(r[0]) != 1u || (r[0] & 0xFFFFFFu) != (c.b1 | (c.b1 << 8)) || rec_pixels(r[0]) != 1u)) ++bad;
            if (c.is_rgb && (rec_class(r[0]) != 2u || (r[0] & 0xFFFFFFu) != (c.rgba & 0xFFFFFFu) || rec_pixels(r[0]) != 1u)) ++bad;
            if (c.is_rgba && (rec_class(r[0]) != 2u || rec_pixels(r[0]) != kRecStash || (r[0] & 0xFFFFFFu) != (c.rgba & 0xFFFFFFu) ||
                              rec_class(r[1]) != 3u || (r[1] & 0xFFu) != (c.rgba >> 24) || rec_pixels(r[1]) != 1u)) ++bad;
        }
    return bad;
}
extern "C" int host_decode_pipeline(const uint8_t* in, int size, uint32_t npx, int och, uint32_t B,
                                    uint8_t* out, long long* stats) {
    return host_decode_pipeline_g(in, size, npx, och, B, 64, out, stats);
}
// len_of / lut_entry against chunk_len / chunk_pixels for all 256 tag bytes; returns mismatches
extern "C" int host_check_lut(void) {
    ChunkLutRef lut; lut.build();
    int bad = 0;
    for (uint32_t b = 0; b < 256; ++b) {
        if (len_of(b) != chunk_len(b) || lut_len(lut.info[b]) != chunk_len(b) || lut_pixels(lut.info[b]) != chunk_pixels(b)) ++bad;
        if (((lut.info[b] & kLutRunBit) != 0u) != crack((unsigned long long)b).is_run) ++bad;
    }
    return bad;
}

static int pipeline(const uint8_t* in, int size, uint32_t npx, int och, uint32_t B, uint32_t GRP,
                    uint8_t* out, long long* stats, int mode) {
    const bool fast = mode >= 1, recmode = mode == 2;
    ChunkLutRef lut; lut.build();
    const uint32_t chunks_end = (uint32_t)size - 8u;
    const uint32_t nseg = (chunks_end - 14u + B - 1u) / B;
    std::vector<ParseRec> parse(nseg);
    std::vector<uint32_t> phase(nseg), px_off(nseg);
    // P1
    for (uint32_t j = 0; j < nseg; ++j) {
        const uint32_t base = 14u + j * B, end = base + B < chunks_end ? base + B : chunks_end;
        if (fast) parse_segment_fast(in, base, end, B, lut, parse[j]); else parse_segment(in, base, end, B, parse[j]);
    }
    // S1, two-level like the kernels: (1) per group and entry phase: exit phase + pixels,
    // (2) chain the groups, (3) chain inside every group from its entry
    const uint32_t ngrp = (nseg + GRP - 1) / GRP;
    std::vector<uint32_t> g_exit(ngrp * 5u), g_phase(ngrp);
    std::vector<unsigned long long> g_px(ngrp * 5u), g_off(ngrp);
    for (uint32_t g = 0; g < ngrp; ++g)
        for (uint32_t ph0 = 0; ph0 < 5; ++ph0) {
            uint32_t ph = ph0; unsigned long long sum = 0;
            for (uint32_t j = g * GRP; j < nseg && j < (g + 1) * GRP; ++j) {
                sum += parse[j].pixels[ph];
                ph = (parse[j].exit_phase >> (3u * ph)) & 7u;
            }
            g_exit[g * 5u + ph0] = ph; g_px[g * 5u + ph0] = sum;
        }
    uint32_t ph = 0; unsigned long long off64 = 0;
    for (uint32_t g = 0; g < ngrp; ++g) {
        g_phase[g] = ph; g_off[g] = off64 > npx ? npx : off64;
        off64 = g_off[g] + g_px[g * 5u + ph];
        ph = g_exit[g * 5u + ph];
    }
    uint32_t off = (uint32_t)(off64 > npx ? npx : off64), n_active = 0;
    for (uint32_t g = 0; g < ngrp; ++g) {
        uint32_t p2 = g_phase[g]; unsigned long long o2 = g_off[g];
        for (uint32_t j = g * GRP; j < nseg && j < (g + 1) * GRP; ++j) {
            phase[j] = p2; px_off[j] = (uint32_t)o2;
            if (o2 < npx && j + 1 > n_active) n_active = j + 1;
            o2 += parse[j].pixels[p2]; if (o2 > npx) o2 = npx;
            p2 = (parse[j].exit_phase >> (3u * p2)) & 7u;
        }
    }
    const uint32_t total_px = off;

    std::vector<uint32_t> entry((size_t)(nseg + 1) * 65u, 0u), fix((size_t)(nseg + 1) * 65u, 0u);
    std::vector<sym_t> summary((size_t)(nseg + 1) * 65u);
    std::vector<SlotRec> srec(nseg);
    std::vector<uint8_t> slot_in(nseg), alpha_in(nseg);
    const uint32_t region = rec_region_dwords(B);              // records of one segment, padded to whole granules
    std::vector<uint32_t> recs(recmode ? (size_t)nseg * region : 0u), rec_gran(nseg, 0u);
    if (nseg) entry[64] = 0xFF000000u;
    uint32_t start = 0, final_px = 0xFF000000u;
    long long rounds = 0, redo = 0, tail_mismatch = 0;
    while (start < n_active) {
        ++rounds;
        const bool refine = fast && rounds > 1;       // later rounds: entry states of the previous round serve as hints
        // P2 + S2
        if (!refine)
        for (uint32_t j = start; j < n_active; ++j) {
            const uint32_t base = 14u + j * B, end = base + B < chunks_end ? base + B : chunks_end;
            if (recmode) rec_gran[j] = transcode_segment(in, base + phase[j], end, lut, &recs[(size_t)j * region], srec[j], &tail_mismatch);
            else if (fast) slot_walk_segment_fast(in, base + phase[j], end, lut, srec[j]); else slot_walk_segment(in, base + phase[j], end, srec[j]);
        }
        if (!refine)
        {   // S2, two-level: compose the transfers of each group, chain groups, apply inside groups
            const uint32_t g0 = start / GRP, g1 = (n_active + GRP - 1) / GRP;
            std::vector<SlotRec> gt(g1);
            for (uint32_t g = g0; g < g1; ++g) {
                SlotRec acc = {0, 1, 0, 0, 0};
                for (uint32_t j = (g * GRP < start ? start : g * GRP); j < n_active && j < (g + 1) * GRP; ++j) acc = slot_compose(acc, srec[j]);
                gt[g] = acc;
            }
            uint32_t slot = hash_px(entry[(size_t)start * 65u + 64u]), alpha = entry[(size_t)start * 65u + 64u] >> 24;
            for (uint32_t g = g0; g < g1; ++g) {
                uint32_t s2 = slot, a2 = alpha;
                for (uint32_t j = (g * GRP < start ? start : g * GRP); j < n_active && j < (g + 1) * GRP; ++j) {
                    slot_in[j] = (uint8_t)s2; alpha_in[j] = (uint8_t)a2;
                    slot_apply(srec[j], s2, a2);
                }
                slot_apply(gt[g], slot, alpha);
            }
        }
        // P3 + S3.  Refinement rounds repeat the pair: a summary made from the hints of the previous round goes stale where the
        // new entry states differ in what the hints are read from, and P4 over every open segment is the expensive part of a round
        // The first round may append such repetitions to its speculative pass as well (QOIMI_DEC_INNER1).
        const int k_inner = getenv("QOIMI_DEC_INNER") ? atoi(getenv("QOIMI_DEC_INNER")) : 4;
        const int k_inner1 = getenv("QOIMI_DEC_INNER1") ? atoi(getenv("QOIMI_DEC_INNER1")) : (recmode && (unsigned long long)chunks_end * 8ull < npx ? 3 : 0);   // as the launcher: flat images only
        const bool round_refine = refine;
        const int inner = round_refine ? (k_inner < 1 ? 1 : k_inner) : 1 + (fast ? k_inner1 : 0);
        for (int it = 0; it < inner; ++it) {
        const bool refine = round_refine || it > 0;
        if (refine) {
            for (uint32_t j = start; j < n_active; ++j) {
                const uint32_t* hs = &entry[(size_t)j * 65u];      // as the kernels: entry states of the previous pass
                const uint32_t epx = hs[64];
                slot_in[j] = (uint8_t)hash_px(epx); alpha_in[j] = (uint8_t)(epx >> 24);
            }
        }
        // P3
        for (uint32_t j = start; j < n_active; ++j) {
            const uint32_t base = 14u + j * B, end = base + B < chunks_end ? base + B : chunks_end;
            SymTab t;
            const uint32_t* ent = &entry[(size_t)j * 65u];
            const uint32_t a_in = alpha_in[j];
            auto hint = [&](uint32_t src) -> uint32_t { return refine ? ent[src] >> 24 : a_in; };
            const sym_t px = recmode ? summarize_records_plain(&recs[(size_t)j * region], rec_gran[j], slot_in[j], alpha_in[j], t, hint, j == 0 || !refine, refine)
                           : fast ? summarize_segment_fast(in, base + phase[j], end, slot_in[j], alpha_in[j], lut, t, hint, j == 0 || !refine)
                                  : summarize_segment(in, base + phase[j], end, slot_in[j], alpha_in[j], t, j == 0 || !refine);
            for (int k = 0; k < 64; ++k) summary[(size_t)j * 65u + k] = t.v[k];
            summary[(size_t)j * 65u + 64u] = px;
        }
        // S3, two-level: compose the symbolic summaries of each group, apply group summaries in
        // sequence to the concrete state, then apply segment summaries inside every group
        {
            const uint32_t g0 = start / GRP, g1 = (n_active + GRP - 1) / GRP;
            std::vector<sym_t> gsum((size_t)g1 * 65u);
            for (uint32_t g = g0; g < g1; ++g) {
                sym_t P[65];
                for (int e = 0; e < 65; ++e) P[e] = sym_make(0u, (uint32_t)e, 0u);     // identity
                for (uint32_t j = (g * GRP < start ? start : g * GRP); j < n_active && j < (g + 1) * GRP; ++j) {
                    sym_t N[65];
                    for (int e = 0; e < 65; ++e) { const sym_t b2 = summary[(size_t)j * 65u + e]; N[e] = sym_compose(b2, P[sym_src(b2)]); }
                    memcpy(P, N, sizeof P);
                }
                memcpy(&gsum[(size_t)g * 65u], P, sizeof P);
            }
            uint32_t cur[65];
            memcpy(cur, &entry[(size_t)start * 65u], 260);
            for (uint32_t g = g0; g < g1; ++g) {
                uint32_t st[65];
                memcpy(st, cur, 260);
                for (uint32_t j = (g * GRP < start ? start : g * GRP); j < n_active && j < (g + 1) * GRP; ++j) {
                    if (j > start) memcpy(&entry[(size_t)j * 65u], st, 260);
                    uint32_t nx[65];
                    for (int e = 0; e < 65; ++e) { const sym_t s3 = summary[(size_t)j * 65u + e]; nx[e] = sym_eval(s3, st[sym_src(s3)]); }
                    memcpy(st, nx, 260);
                }
                uint32_t nc[65];
                for (int e = 0; e < 65; ++e) { const sym_t s3 = gsum[(size_t)g * 65u + e]; nc[e] = sym_eval(s3, cur[sym_src(s3)]); }
                memcpy(cur, nc, 260);
            }
        }
        }
        // P4 + check
        uint32_t first_bad = 0xFFFFFFFFu;
        for (uint32_t j = start; j < n_active; ++j) {
            const uint32_t base = 14u + j * B, end = base + B < chunks_end ? base + B : chunks_end;
            Tab32 t;
            memcpy(t.v, &entry[(size_t)j * 65u], 256);
            uint32_t px = entry[(size_t)j * 65u + 64u];
            if (recmode) px = och == 4 ? decode_records<4>(&recs[(size_t)j * region], rec_gran[j], px, t, out, px_off[j], npx)
                                       : decode_records<3>(&recs[(size_t)j * region], rec_gran[j], px, t, out, px_off[j], npx);
            else if (fast) px = och == 4 ? decode_segment_fast<4>(in, base + phase[j], end, px, lut, t, out, px_off[j], npx)
                                    : decode_segment_fast<3>(in, base + phase[j], end, px, lut, t, out, px_off[j], npx);
            else px = och == 4 ? decode_segment<4>(in, base + phase[j], end, px, t, out, px_off[j], npx)
                               : decode_segment<3>(in, base + phase[j], end, px, t, out, px_off[j], npx);
            if (j + 1 < n_active) {
                const uint32_t* nxt = &entry[(size_t)(j + 1) * 65u];
                if (nxt[64] != px || memcmp(nxt, t.v, 256) != 0) {
                    memcpy(&fix[(size_t)(j + 1) * 65u], t.v, 256);
                    fix[(size_t)(j + 1) * 65u + 64u] = px;
                    if (j + 1 < first_bad) first_bad = j + 1;
                }
            } else {
                final_px = px;
            }
        }
        if (getenv("QOIMI_HOST_TRACE")) fprintf(stderr, "round %lld start %u first_bad %u\n", rounds, start, first_bad);
        if (first_bad == 0xFFFFFFFFu) break;
        memcpy(&entry[(size_t)first_bad * 65u], &fix[(size_t)first_bad * 65u], 260);
        redo += n_active - first_bad;
        start = first_bad;
    }
    // fill
    const uint32_t fpx = n_active ? final_px : 0xFF000000u;
    for (uint32_t i = total_px; i < npx; ++i) {
        if (och == 4) reinterpret_cast<uint32_t*>(out)[i] = fpx;
        else { out[(size_t)i * 3] = (uint8_t)fpx; out[(size_t)i * 3 + 1] = (uint8_t)(fpx >> 8); out[(size_t)i * 3 + 2] = (uint8_t)(fpx >> 16); }
    }
    if (stats) { stats[0] = rounds; stats[1] = redo; stats[2] = nseg; stats[3] = tail_mismatch; }
    return 0;
}
