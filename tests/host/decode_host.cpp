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
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../qoi_amd/csrc/qoi_decode_core.h"

using namespace qoimi;

namespace {
struct SymTab { sym_t v[64]; sym_t get(uint32_t k) const { return v[k]; } void set(uint32_t k, sym_t x) { v[k] = x; } };
struct Tab32 { uint32_t v[64]; uint32_t get(uint32_t k) const { return v[k]; } void set(uint32_t k, uint32_t x) { v[k] = x; } };
}

// returns 0; stats[0] = rounds, stats[1] = segments re-decoded, stats[2] = segments
extern "C" int host_decode_pipeline(const uint8_t* in, int size, uint32_t npx, int och, uint32_t B,
                                    uint8_t* out, long long* stats) {
    const uint32_t chunks_end = (uint32_t)size - 8u;
    const uint32_t nseg = (chunks_end - 14u + B - 1u) / B;
    std::vector<ParseRec> parse(nseg);
    std::vector<uint32_t> phase(nseg), px_off(nseg);
    // P1
    for (uint32_t j = 0; j < nseg; ++j) {
        const uint32_t base = 14u + j * B, end = base + B < chunks_end ? base + B : chunks_end;
        parse_segment(in, base, end, B, parse[j]);
    }
    // S1
    uint32_t ph = 0, off = 0, n_active = 0;
    for (uint32_t j = 0; j < nseg; ++j) {
        phase[j] = ph; px_off[j] = off;
        if (off < npx) n_active = j + 1;
        const unsigned long long t = (unsigned long long)off + parse[j].pixels[ph];
        ph = (parse[j].exit_phase >> (3u * ph)) & 7u;
        off = t > npx ? npx : (uint32_t)t;
    }
    const uint32_t total_px = off;

    std::vector<uint32_t> entry((size_t)(nseg + 1) * 65u, 0u), fix((size_t)(nseg + 1) * 65u, 0u);
    std::vector<sym_t> summary((size_t)(nseg + 1) * 65u);
    std::vector<SlotRec> srec(nseg);
    std::vector<uint8_t> slot_in(nseg), alpha_in(nseg);
    if (nseg) entry[64] = 0xFF000000u;
    uint32_t start = 0, final_px = 0xFF000000u;
    long long rounds = 0, redo = 0;
    while (start < n_active) {
        ++rounds;
        // P2 + S2
        for (uint32_t j = start; j < n_active; ++j) {
            const uint32_t base = 14u + j * B, end = base + B < chunks_end ? base + B : chunks_end;
            slot_walk_segment(in, base + phase[j], end, srec[j]);
        }
        uint32_t slot = hash_px(entry[(size_t)start * 65u + 64u]), alpha = entry[(size_t)start * 65u + 64u] >> 24;
        for (uint32_t j = start; j < n_active; ++j) {
            slot_in[j] = (uint8_t)slot; alpha_in[j] = (uint8_t)alpha;
            slot_apply(srec[j], slot, alpha);
        }
        // P3
        for (uint32_t j = start; j < n_active; ++j) {
            const uint32_t base = 14u + j * B, end = base + B < chunks_end ? base + B : chunks_end;
            SymTab t;
            const sym_t px = summarize_segment(in, base + phase[j], end, slot_in[j], alpha_in[j], t);
            for (int k = 0; k < 64; ++k) summary[(size_t)j * 65u + k] = t.v[k];
            summary[(size_t)j * 65u + 64u] = px;
        }
        // S3
        for (uint32_t j = start; j + 1 < n_active; ++j) {
            const uint32_t* cur = &entry[(size_t)j * 65u];
            uint32_t* nxt = &entry[(size_t)(j + 1) * 65u];
            for (int e = 0; e < 65; ++e) {
                const sym_t s = summary[(size_t)j * 65u + e];
                nxt[e] = sym_eval(s, cur[sym_src(s)]);
            }
        }
        // P4 + check
        uint32_t first_bad = 0xFFFFFFFFu;
        for (uint32_t j = start; j < n_active; ++j) {
            const uint32_t base = 14u + j * B, end = base + B < chunks_end ? base + B : chunks_end;
            Tab32 t;
            memcpy(t.v, &entry[(size_t)j * 65u], 256);
            uint32_t px = entry[(size_t)j * 65u + 64u];
            px = och == 4 ? decode_segment<4>(in, base + phase[j], end, px, t, out, px_off[j], npx)
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
    if (stats) { stats[0] = rounds; stats[1] = redo; stats[2] = nseg; }
    return 0;
}
