// qoi_decode.hip — MI355X-native QOI decoder kernels (gfx950, wave64).
//
// Replaces the sequential loop of the reference decoder (qoi.h:488-590).  See
// qoi_decode_core.h for the scheme (P1 parse, P2 slot walk, P3 symbolic summary,
// P4 genuine decode + exit-state check).  This file holds the grid plumbing:
// one lane per stream segment for the P-passes, one wavefront per image for the
// chaining passes (S1..S3), plus the fill of pixels a truncated stream never reaches
// (Appendix B item 1: they repeat the last pixel).
#include "qoi_dev.h"
#include "qoi_kernels.h"
#include "qoi_decode_core.h"

namespace qoimi {

// locate (image, segment-in-image) of global segment q: images are few thousand at most
__device__ __forceinline__ uint32_t find_image(const DecImage* __restrict__ im, uint32_t n_images, uint32_t q) {
    uint32_t lo = 0, hi = n_images;            // invariant: seg_base[lo] <= q < seg_base[hi]
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (im[mid].seg_base <= q) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------
// P1: parse summaries (lane = segment)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_parse(DecParams p) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= p.total_segs) return;
    const uint32_t img = find_image(p.images, p.n_images, q);
    const DecImage im = p.images[img];
    const uint32_t j = q - im.seg_base;
    const uint32_t base = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t end = min(base + p.seg_bytes, im.chunks_end);
    ParseRec r;
    parse_segment(p.streams + im.stream_off, base, end, p.seg_bytes, r);
    p.parse[q] = r;
}

// ---------------------------------------------------------------------------------
// S1: chain the parse summaries of one image (wavefront per image): entry phase and
// pixel offset of every segment; pixel total.  Exact.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void dec_chain_parse(DecParams p) {
    const uint32_t img = blockIdx.x, lane = lane_id();
    DecImage im = p.images[img];
    uint32_t phase = 0, off = 0;
    uint32_t n_active = 0;                       // segments that start before the pixel limit
    for (uint32_t j0 = 0; j0 < im.nseg; j0 += 64u) {
        const uint32_t j = j0 + lane;
        ParseRec r; r.exit_phase = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) r.pixels[k] = 0;
        if (j < im.nseg) r = p.parse[im.seg_base + j];
        uint32_t my_phase = 0, my_off = 0;
        const uint32_t cnt = min(64u, im.nseg - j0);
        for (uint32_t l = 0; l < cnt; ++l) {
            if (lane == l) { my_phase = phase; my_off = off; }
            if (off < im.npx) n_active = j0 + l + 1u;
            // select by the (uniform) phase, then broadcast lane l's numbers
            const uint32_t sel_px = phase == 0 ? r.pixels[0] : phase == 1 ? r.pixels[1] : phase == 2 ? r.pixels[2]
                                   : phase == 3 ? r.pixels[3] : r.pixels[4];
            const uint32_t sel_ex = (r.exit_phase >> (3u * phase)) & 7u;
            const uint32_t add = read_lane_dyn(sel_px, l);
            phase = read_lane_dyn(sel_ex, l);
            const u64 t = (u64)off + add;
            off = t > im.npx ? im.npx : (uint32_t)t;
        }
        if (j < im.nseg) { p.entry_phase[im.seg_base + j] = (uint8_t)my_phase; p.px_off[im.seg_base + j] = my_off; }
    }
    if (lane == 0) {
        p.images[img].total_px = off;
        p.images[img].n_active = n_active;
        p.images[img].start_seg = 0;
        p.first_bad[img] = 0xFFFFFFFFu;
    }
}

// ---------------------------------------------------------------------------------
// P2: speculative slot/alpha transfer (lane = segment)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dec_slot_walk(DecParams p) {
    const uint32_t q = blockIdx.x * 256u + threadIdx.x;
    if (q >= p.total_segs) return;
    const uint32_t img = find_image(p.images, p.n_images, q);
    const DecImage im = p.images[img];
    const uint32_t j = q - im.seg_base;
    if (j < im.start_seg || j >= im.n_active) return;
    const uint32_t base = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t end = min(base + p.seg_bytes, im.chunks_end);
    SlotRec r;
    slot_walk_segment(p.streams + im.stream_off, base + p.entry_phase[q], end, r);
    p.slot_rec[q] = r;
}

// S2: chain slot/alpha over the active segments of one image from start_seg.
// The chain's start value is the hash/alpha of the concrete entry pixel of start_seg.
__global__ __launch_bounds__(64) void dec_chain_slots(DecParams p) {
    const uint32_t img = blockIdx.x, lane = lane_id();
    const DecImage im = p.images[img];
    if (im.start_seg >= im.n_active) return;
    const uint32_t px0 = p.entry[(size_t)(im.seg_base + im.start_seg) * 65u + 64u];
    uint32_t slot = hash_px(px0), alpha = px0 >> 24;
    for (uint32_t j0 = im.start_seg; j0 < im.n_active; j0 += 64u) {
        const uint32_t j = j0 + lane;
        SlotRec r = {0, 1, 0, 0, 0};
        if (j < im.n_active) r = p.slot_rec[im.seg_base + j];
        const uint32_t packed = r.hc | (r.h_rel << 8) | (r.h_alpha << 9) | (r.a_abs << 10) | ((uint32_t)r.ac << 16);
        uint32_t my_slot = 0, my_alpha = 0;
        const uint32_t cnt = min(64u, im.n_active - j0);
        for (uint32_t l = 0; l < cnt; ++l) {
            if (lane == l) { my_slot = slot; my_alpha = alpha; }
            const uint32_t w = read_lane_dyn(packed, l);
            SlotRec t; t.hc = w & 63u; t.h_rel = (w >> 8) & 1u; t.h_alpha = (w >> 9) & 1u; t.a_abs = (w >> 10) & 1u; t.ac = (w >> 16) & 0xFFu;
            slot_apply(t, slot, alpha);
        }
        if (j < im.n_active) { p.slot_in[im.seg_base + j] = (uint8_t)my_slot; p.alpha_in[im.seg_base + j] = (uint8_t)my_alpha; }
    }
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
    const uint32_t lane = lane_id();
    const uint32_t q = blockIdx.x * 64u + lane;
    if (q >= p.total_segs) return;
    const uint32_t img = find_image(p.images, p.n_images, q);
    const DecImage im = p.images[img];
    const uint32_t j = q - im.seg_base;
    if (j < im.start_seg || j >= im.n_active) return;
    const uint32_t base = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t end = min(base + p.seg_bytes, im.chunks_end);
    LdsSymTab tab{&s_tab[lane]};
    const sym_t px = summarize_segment(p.streams + im.stream_off, base + p.entry_phase[q], end,
                                       p.slot_in[q], p.alpha_in[q], tab);
    sym_t* dst = p.summary + (size_t)q * 65u;
    for (uint32_t k = 0; k < 64u; ++k) dst[k] = tab.get(k);
    dst[64] = px;
}

// S3: apply the summaries in sequence to the concrete state (wavefront per image,
// lane = table slot; the running pixel is kept redundantly by every lane).
__global__ __launch_bounds__(64) void dec_chain_state(DecParams p) {
    const uint32_t img = blockIdx.x, lane = lane_id();
    const DecImage im = p.images[img];
    if (im.start_seg >= im.n_active) return;
    size_t q = (size_t)im.seg_base + im.start_seg;
    uint32_t tabv = p.entry[q * 65u + lane];        // concrete entry state of start_seg is given
    uint32_t pxv = p.entry[q * 65u + 64u];
    const uint32_t last = im.n_active - 1u;
    sym_t s_tab = 0, s_px = 0;
    if (im.start_seg < last) { s_tab = p.summary[q * 65u + lane]; s_px = p.summary[q * 65u + 64u]; }
    for (uint32_t j = im.start_seg; j < last; ++j, ++q) {
        const sym_t c_tab = s_tab, c_px = s_px;
        if (j + 1u < last) { s_tab = p.summary[(q + 1u) * 65u + lane]; s_px = p.summary[(q + 1u) * 65u + 64u]; }   // prefetch
        const uint32_t src_t = sym_src(c_tab), src_p = sym_src(c_px);
        const uint32_t g_t = gather_lane(tabv, src_t & 63u), g_p = gather_lane(tabv, src_p & 63u);
        const uint32_t ntab = sym_eval(c_tab, src_t == 64u ? pxv : g_t);
        const uint32_t npx = sym_eval(c_px, src_p == 64u ? pxv : g_p);
        tabv = ntab; pxv = npx;
        p.entry[(q + 1u) * 65u + lane] = tabv;
        if (lane == 0) p.entry[(q + 1u) * 65u + 64u] = pxv;
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
    const uint32_t lane = lane_id();
    const uint32_t q = blockIdx.x * 64u + lane;
    if (q >= p.total_segs) return;
    const uint32_t img = find_image(p.images, p.n_images, q);
    const DecImage im = p.images[img];
    const uint32_t j = q - im.seg_base;
    if (j < im.start_seg || j >= im.n_active) return;
    const uint32_t base = (uint32_t)kHeaderBytes + j * p.seg_bytes;
    const uint32_t end = min(base + p.seg_bytes, im.chunks_end);
    LdsTab32 tab{&s_tab[lane]};
    const uint32_t* __restrict__ ent = p.entry + (size_t)q * 65u;
    for (uint32_t k = 0; k < 64u; ++k) tab.set(k, ent[k]);
    uint32_t px = ent[64];
    px = decode_segment<OCH>(p.streams + im.stream_off, base + p.entry_phase[q], end, px, tab,
                             p.pixels + (size_t)img * p.pixel_stride, p.px_off[q], im.npx);
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
    }
    hipLaunchKernelGGL(dec_chain_parse, dim3(p.n_images), dim3(64), 0, st, p);
    hipLaunchKernelGGL(dec_init_state, dim3(p.n_images), dim3(64), 0, st, p);
    tm->mark(kT_dec_chain_parse, st);
}

void launch_decode_round(const DecParams& p, int out_channels, hipStream_t st, KernelTimer* tm) {
    if (!p.total_segs) return;
    const uint32_t b256 = (p.total_segs + 255u) / 256u, b64 = (p.total_segs + 63u) / 64u;
    tm->mark(kT_begin, st);
    hipLaunchKernelGGL(dec_slot_walk, dim3(b256), dim3(256), 0, st, p);
    tm->mark(kT_dec_slot_walk, st);
    hipLaunchKernelGGL(dec_chain_slots, dim3(p.n_images), dim3(64), 0, st, p);
    tm->mark(kT_dec_chain_slots, st);
    hipLaunchKernelGGL(dec_summarize, dim3(b64), dim3(64), 0, st, p);
    tm->mark(kT_dec_summarize, st);
    hipLaunchKernelGGL(dec_chain_state, dim3(p.n_images), dim3(64), 0, st, p);
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
