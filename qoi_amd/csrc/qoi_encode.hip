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

template <int CH, int K>
__global__ __launch_bounds__(256) void enc_slabs(EncParams p) {
    __shared__ uint32_t s_table[4][64];
    __shared__ u64 s_mask[4][64];
    __shared__ uint32_t s_ticket;

    const uint32_t wave = threadIdx.x >> 6, lane = lane_id();
    // Slab ids are handed out in START order so that every predecessor a look-back can
    // wait on is already running or finished (no reliance on dispatch order; guide G16).
    if (threadIdx.x == 0) s_ticket = atomicAdd(p.ticket, 1u);
    __syncthreads();
    const uint32_t g = s_ticket * 4u + wave;
    const uint32_t total = p.n_images * p.spi;
    if (g >= total) return;
    const uint32_t img = g / p.spi, s = g - img * p.spi;
    const uint8_t* __restrict__ pix = p.pixels + (size_t)img * p.pixel_stride;
    const uint32_t n = p.npx, lo = s * (64u * K);

    // ---- entry state: colour table + last edge position ------------------------------
    {
        const uint32_t G = img * p.gpi + (s >> 6);
        const uint32_t loc = p.ent_tab[(size_t)g * 64u + lane];
        const uint32_t far = p.gent_tab[(size_t)G * 64u + lane];
        const u64 lv = p.ent_valid[g];
        s_table[wave][lane] = ((lv >> lane) & 1ull) ? loc : far;
        s_mask[wave][lane] = 0;
    }
    int last_edge = max(p.ent_le[g], p.gent_le[img * p.gpi + (s >> 6)]);   // max edge position < lo, or -1
    uint32_t carry = (lo > 0) ? load_px<CH>(pix, lo - 1) : kInitPx;
    __builtin_amdgcn_wave_barrier();

    const u64 lane_bit = 1ull << lane;
    const u64 below = lane_bit - 1ull;

    // ---- pass 1: classify every pixel of the slab, keep chunks in registers ------------
    uint32_t px_reg[K];
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const uint32_t i = lo + t * 64u + lane;
        px_reg[t] = (i < n) ? load_px<CH>(pix, i) : 0u;
    }
    u64 enc[K];             // bits 0..47: up to 6 chunk bytes in emission order; bits 56..58: length
    uint32_t lane_bytes = 0;
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const uint32_t base = lo + t * 64u;
        const uint32_t i = base + lane;
        const bool inb = i < n;
        const uint32_t px = px_reg[t];
        const uint32_t prev = from_lane_below(px, carry);
        carry = read_lane(px, 63);
        const bool edge = inb && px != prev;
        const u64 E = __ballot(edge);

        // last edge strictly before this pixel
        const u64 eb = E & below;
        const int le = eb ? (int)base + msb64(eb) : last_edge;
        if (E) last_edge = (int)base + msb64(E);

        // ---- colour-table probe (qoi.h:430-436) for edge pixels ------------------------
        // Lanes of this step that share a slot: every edge lane ORs its bit into the slot's
        // 64-bit LDS word; the word then lists all of them (order-independent).
        const uint32_t so = slot_byte_offset(px);
        volatile u64* mword = &s_mask[wave][so >> 2];
        volatile uint32_t* tword = &s_table[wave][so >> 2];
        if (edge) atomicOr((u64*)mword, lane_bit);
        __builtin_amdgcn_wave_barrier();
        u64 same = 0; uint32_t tval = 0;
        if (edge) { same = *mword; tval = *tword; }
        __builtin_amdgcn_wave_barrier();
        if (edge) *mword = 0;
        const u64 pred = same & below;
        // table content seen by this pixel = nearest earlier edge lane with the same slot, else
        // the table carried in from earlier steps/slabs
        const uint32_t pv = gather_lane(px, pred ? (uint32_t)msb64(pred) : lane);
        const uint32_t seen = pred ? pv : tval;
        const bool hit = edge && seen == px;
        // the last edge lane of each slot leaves its pixel in the table for later steps
        if (edge && ((same >> lane) >> 1) == 0) *tword = px;
        __builtin_amdgcn_wave_barrier();

        // ---- chunk bytes --------------------------------------------------------------
        u64 bytes = 0; uint32_t len = 0;
        if (edge) {
            u64 lb; uint32_t ll;
            literal_chunk(px, prev, lb, ll);
            if (hit) { lb = kTagIndex | (so >> 2); ll = 1; }
            const uint32_t pend = (uint32_t)((int)i - 1 - le) % 62u;       // repeats not yet flushed (qoi.h:425-428)
            if (pend) { bytes = (lb << 8) | (kTagRun | (pend - 1u)); len = ll + 1u; }
            else { bytes = lb; len = ll; }
        } else if (inb) {
            const uint32_t r = (uint32_t)((int)i - le);                    // repeats ending here
            const uint32_t q = r % 62u;
            if (q == 0u || i == n - 1u) {                                  // qoi.h:417
                bytes = kTagRun | (q == 0u ? 61u : q - 1u); len = 1;
            }
        }
        enc[t] = bytes | ((u64)len << 56);
        lane_bytes += len;
    }

    // ---- slab byte count and its offset: decoupled look-back over earlier slabs ---------
    const uint32_t slab_bytes = wave_sum(lane_bytes);
    u64 excl = 0;
    {
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

    // ---- pass 2: emit -------------------------------------------------------------------
    uint8_t* __restrict__ out = p.out + (size_t)img * p.out_stride;
    if (s == 0 && lane < (uint32_t)kHeaderBytes) {        // 14-byte header (qoi.h:384-388)
        const uint32_t w = p.width, h = p.height;
        uint8_t b;
        switch (lane) {
            case 0: b = 'q'; break; case 1: b = 'o'; break; case 2: b = 'i'; break; case 3: b = 'f'; break;
            case 4: b = w >> 24; break; case 5: b = w >> 16; break; case 6: b = w >> 8; break; case 7: b = w; break;
            case 8: b = h >> 24; break; case 9: b = h >> 16; break; case 10: b = h >> 8; break; case 11: b = h; break;
            case 12: b = p.channels; break; default: b = p.colorspace; break;
        }
        out[lane] = b;
    }
    u64 pos = (u64)kHeaderBytes + excl;
#pragma unroll
    for (int t = 0; t < K; ++t) {
        const u64 e = enc[t];
        const uint32_t len = (uint32_t)(e >> 56);
        const u64 b0 = __ballot(len & 1u), b1 = __ballot(len & 2u), b2 = __ballot(len & 4u);
        const uint32_t off = count_below(b0) + 2u * count_below(b1) + 4u * count_below(b2);
        uint8_t* dst = out + pos + off;
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            if (__ballot(len > (uint32_t)k) == 0) break;
            if (len > (uint32_t)k) dst[k] = (uint8_t)(e >> (8 * k));
        }
        pos += (u64)__builtin_popcountll(b0) + 2u * (u64)__builtin_popcountll(b1) + 4u * (u64)__builtin_popcountll(b2);
    }
    if (s == p.spi - 1u) {                                // trailer (qoi.h:339,480-482) + *out_len
        if (lane < (uint32_t)kTrailerBytes) out[pos + lane] = (lane == 7u) ? 1 : 0;
        if (lane == 0) p.out_len[img] = (int)(pos + kTrailerBytes);
    }
}

// ---------------------------------------------------------------------------------
// host-side launcher
// ---------------------------------------------------------------------------------
template <int CH, int K>
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
    hipLaunchKernelGGL((enc_slabs<CH, K>), dim3(blocks), dim3(256), 0, st, p);
    tm->mark(kT_enc_slabs, st);
}

void launch_encode(const EncParams& p, hipStream_t st, KernelTimer* tm) {
    if (p.channels == 4) launch_encode_t<4, kEncSteps>(p, st, tm);
    else launch_encode_t<3, kEncSteps>(p, st, tm);
}

}  // namespace qoimi
