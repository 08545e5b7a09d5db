// qoi_kernels.h — parameter blocks and launchers shared by the kernels and the C-ABI shim.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace qoimi {

typedef unsigned long long u64;

constexpr int kHeaderBytes = 14, kTrailerBytes = 8;   // qoi.h:326, qoi.h:339

// Optional per-kernel timing with HIP events recorded on the launch stream (bench.py's
// roofline figure).  mark(tag) closes the interval of the kernel launched just before it.
enum KernelTag { kT_begin = 0,
                 kT_enc_summary, kT_enc_scan_groups, kT_enc_scan_images, kT_enc_slabs, kT_enc_slabs_generic, kT_enc_offsets, kT_enc_compact,
                 kT_dec_parse, kT_dec_chain_parse, kT_dec_slot_walk, kT_dec_chain_slots, kT_dec_summarize,
                 kT_dec_chain_state, kT_dec_segments, kT_dec_restart, kT_dec_fill, kT_dec_expand,
                 kT_enc_total, kT_dec_total,    // a whole qoimi_encode_batch / qoimi_decode_batch on the caller's stream (kernels of a call may overlap)
                 kT_count };
struct KernelTimer {
    static constexpr int kMax = 512;
    hipEvent_t ev[kMax]; int tag[kMax]; int n = 0; bool on = false; bool created = false;
    void mark(int t, hipStream_t st) {
        if (!on || n >= kMax) return;
        tag[n] = t; (void)hipEventRecord(ev[n], st); ++n;
    }
};

// ---- encode ------------------------------------------------------------------------
constexpr int kEncSteps = 16;                         // 64-pixel steps per slab
constexpr uint32_t kEncSlabPx = 64u * kEncSteps;      // pixels per slab: the unit of the generic entry-state passes
constexpr uint32_t kEncMaxSetSlabs = 16;              // a wavefront encodes a SET of 1..16 consecutive slabs
constexpr uint32_t kEncSlabWorst = kEncSlabPx * 5u;   // most bytes a slab can produce (QOI_OP_RGBA everywhere)
constexpr uint32_t kEncPoolSlots = 8192;              // look-back mode: scratch slots of the sets that spill, handed out by a bitmap (more than the 6144
                                                      // wavefronts of enc_sets a chip holds at a time: a set keeps its slot from its first spill to its copy-out)
constexpr uint32_t kEncPoolMapStride = 16;            // u64 words between two words of the pool's bitmap: one word per 128-byte line
constexpr uint32_t kEncTreeMaxSets = 12288;          // automatic choice of the placement: an image of more sets than this is placed order-free, not by the tree
constexpr uint32_t kG2TailGroups = 2;                 // state look-back: groups (of 512 pixels) at a set's end that are walked first
constexpr uint32_t kEncGenSetSlabs = 8;               // slabs per set of the generic pass when it places by look-back (flat content: few bytes per slab)

// Calls of differently shaped images (qoimi_encode_images): everything the kernels take from the call's one shape otherwise, per image.
// Such calls are placed order-free (every set parks its bytes in a slot of its own, enc_offsets + enc_compact): no set waits for another,
// so any mix of sizes will do.
struct EncImage {
    size_t   pixel_off, out_off;   // image at pixels + pixel_off, stream at out + out_off
    uint32_t npx, spi, gpi, sets;  // pixels, slabs, 64-slab groups, sets
    uint32_t set_base, slab_base, grp_base, unit_base;   // global index of the image's first set / slab / group / (four-set) work unit
    uint32_t width, height, colorspace, len_index;   // len_index: where the image's stream length goes in out_len (the caller's image number)
};

struct EncParams {
    const EncImage* img_tab; // device [n_images + 1] (the last entry holds the totals in its *_base fields); nullptr: one shape (the fields below)
    uint32_t total_slabs;    // set by the launcher: slabs of all images
    const uint8_t* pixels;   // image i at pixels + i*pixel_stride
    size_t pixel_stride;
    uint32_t npx;            // width*height
    uint32_t n_images;
    uint32_t spi;            // slabs per image
    uint32_t gpi;            // 64-slab groups per image
    uint32_t set_slabs;      // slabs per set (R)
    uint32_t set_px;         // pixels per set = R * kEncSlabPx
    uint32_t sets_per_image; // ceil(spi / R)
    uint32_t set_stride;     // bytes of a set's scratch slot (R * kEncSlabWorst + 16)
    uint32_t width, height;
    uint8_t channels, colorspace;
    uint8_t probe_xchg;      // 1: ds_wrxchg colour-table probe (needs the LDS order self-test to have passed)
    uint8_t use_ticket;      // 1: set ids by atomic ticket (start order); 0: by blockIdx
    uint8_t lookback;        // 1: a set finds its place in the stream by decoupled look-back and writes its bytes itself;
                             // 2: the same by the three-level tree of byte counts (tree1 / tree2; calls of a few large images);
                             // 0: order-free - sets park their bytes in scratch slots, enc_offsets + enc_compact place them
    uint8_t warm;            // 1: sets find their entry state themselves (look-back window), E1/E2 only for flagged images
    uint8_t only_flagged;    // set by the launcher: this pass handles images with need_generic[img] != 0 only
    uint32_t n_units;        // set by the launcher: (image, four consecutive sets) work units
    uint32_t spread;         // 1 (default): the wavefronts of a workgroup serve consecutive images (env QOIMI_ENC_SPREAD=0: all four take tickets of one image)
    uint32_t gen_grid_div;   // 0: the pass over flagged images runs with the small grid; N: with 1/N of its units (the previous batch of the context held flagged images)
    uint32_t gen_small_div;  // 0 (= 32): the passes that usually find nothing run with 1/N of the full grid, 2048 workgroups at least (env QOIMI_ENC_GEN_GRID_DIV under QOIMI_TUNING)
    uint32_t gen_slabs;      // slabs per set of the pass over flagged images (kEncGenSetSlabs; env QOIMI_ENC_GEN_SLABS)
    uint32_t spin_bound;     // polls a placement wait makes before it gives up (err bit 0): 2^22 with tickets (start order), 2^15 for the tree by workgroup index
    uint32_t uni;            // 1 (env QOIMI_ENC_UNI=1, or a small call behind one that held flat stretches): one pass (enc_sets<ENTRY 3>) - sets whose look-back window does not do take the state look-back themselves; g2_rec holds a record per set of the FIRST pass
    uint32_t all_g2;         // 1: no first pass, every image of the call goes through the pass over flagged images (the context's previous batch held flagged images only)
    uint32_t pipe;           // 1 (env QOIMI_ENC_PIPE, with persist): a wavefront asks for its next set's first loads in front of its current set's placement
    uint32_t persist;        // 0: one workgroup per unit; else the first pass runs at most this many workgroups (env QOIMI_ENC_PERSIST, a test knob)
    // workspace
    uint32_t* sum_tab;   u64* sum_valid;  int* sum_le;     // E1 out        [n_images*spi]
    uint32_t* ent_tab;   u64* ent_valid;  int* ent_le;     // E2a out       [n_images*spi]
    uint32_t* grp_tab;   u64* grp_valid;  int* grp_le;     // E2a aggregate [n_images*gpi]
    uint32_t* gent_tab;  int* gent_le;                     // E2b out       [n_images*gpi]
    u64* status;         // look-back records [n_images*sets_per_image]   -- zeroed before every launch
    u64* tree1;          // lookback == 2: totals of the groups of 64 sets [n_images * ceil(sets_per_image / 64)]        -- zeroed before every launch
    u64* tree2;          // lookback == 2: totals of the blocks of 64 groups [n_images * ceil(sets_per_image / 4096)]   -- zeroed before every launch
    uint32_t* ticket;    // per-image ticket counters [n_images]          -- zeroed before every launch
    uint32_t* err;       // liveness-bound flag               -- zeroed before every launch
    uint32_t* need_generic;  // [n_images] image needs the E1/E2 path  -- zeroed before every launch
    uint32_t* any_generic;   // [1]                                    -- zeroed before every launch
    uint32_t* zero_next;     // tree placement (may be null): the records / tickets / flags / pool map of the context's NEXT such call - this call's first
    uint32_t zero_next_dwords;   // enc_sets launch zeroes them in passing (two such regions, used in turn), and that call needs no hipMemsetAsync
    uint32_t* host_hint;     // pinned HOST word (may be null): the first set of a call whose look-back window does not do leaves the call's number there - the next small call's choice of pass
    uint8_t* scratch;    // order-free mode: [n_images*sets_per_image][set_stride] parked sets; look-back mode (pool = 1): [pool_slots + 1][set_stride],
                         // the spilled pieces of the sets that hold a slot (the last slot is the emergency slot of an exhausted pool: err bit 1)
    u64* pool_map;       // [pool_slots / 64 * kEncPoolMapStride] look-back mode: bit set = slot taken      -- zeroed before every launch
    uint32_t pool_slots; // multiple of 64
    uint8_t pool;        // 1: scratch slots come from the pool (look-back mode)
    // the generic pass of a look-back call places by look-back too, with its own records / tickets and kEncGenSetSlabs slabs per set
    u64* status_gen;     // [n_images * ceil(spi / kEncGenSetSlabs)]                                     -- zeroed before every launch
    uint32_t* ticket_gen;    // [n_images]                                                             -- zeroed before every launch
    u64* tree1_gen; u64* tree2_gen;   // lookback == 2: the generic pass's group / block totals                -- zeroed before every launch
    u64* g2_rec;         // ENTRY 2 (flagged images by state look-back over their sets): [n_images * ceil(spi / kEncGenSetSlabs)][65] granules tagged
                         // with `epoch` - never zeroed between calls; nullptr: the summary passes (enc_slab_summary + scans + ENTRY 0)
    uint32_t epoch;      // the context's encode call number (29 bits are compared)
    uint32_t* set_size;  // [n_images*sets_per_image]  order-free mode
    uint32_t* set_off;   // [n_images*sets_per_image]  order-free mode
    // output
    uint8_t* out; size_t out_stride; int* out_len;
};

// what of an encode a call launches: the slab passes (pixels -> parked slab bytes + sizes), the placement passes
// (exclusive scan of the sizes + compaction), or both back to back
enum EncPhase { kEncSlabs = 1, kEncPlace = 2, kEncAll = 3 };
void launch_encode(const EncParams& p, hipStream_t st, KernelTimer* tm, int phases = kEncAll);
void launch_encode_mixed(const EncParams& p, uint32_t units, uint32_t slabs, uint32_t groups, uint32_t sets, hipStream_t st, KernelTimer* tm);
int run_lds_order_selftest(hipStream_t st);
void launch_lds_order_selftest(uint32_t* d_out, hipStream_t st);

// ---- decode ------------------------------------------------------------------------
struct ParseRec; struct SlotRec;

struct DecImage {
    size_t   stream_off;   // stream i starts at streams + stream_off
    uint32_t chunks_end;   // size - 8 (qoi.h:539); chunk region is [14, chunks_end)
    uint32_t npx;          // width*height
    uint32_t seg_base;     // global index of this image's first segment
    uint32_t nseg;         // ceil((chunks_end-14)/seg_bytes)
    uint32_t grp_base;     // global index of this image's first 64-segment group
    uint32_t ngrp;         // ceil(nseg/64)
    // filled on the device
    uint32_t total_px;     // pixels produced by all chunks, clamped to npx        (S1)
    uint32_t n_active;     // segments that start before the pixel limit            (S1)
    uint32_t start_seg;    // first segment still to be (re)decoded; n_active: done
    uint32_t final_px;     // exit pixel of the last active segment                 (P4)
    uint32_t desc_base;    // flat images (run descriptors): index of the image's first segment among the flat images' segments; kNoRunDesc: none
    uint32_t out_index;    // the image's place in the caller's pixel buffer: pixels + out_index * pixel_stride (a call decoded class by class: not the table index)
};
constexpr uint32_t kNoRunDesc = 0xFFFFFFFFu;

// "Flat" images - a stream of less than a byte per eight pixels (UI frames, constant frames): a pass over their chunks costs a
// fraction of a pass over their pixels, so the first round spends a few refinement passes of P3 + S3 on them before its P4.
inline __host__ __device__ bool dec_image_is_flat(uint32_t chunks_end, uint32_t npx) { return (unsigned long long)chunks_end * 8ull < (unsigned long long)npx; }

struct DecParams {
    const uint8_t* streams;
    DecImage* images;      // device array [n_images]
    uint32_t n_images, total_segs, total_grps, seg_bytes;
    uint32_t fine_per_seg, fine_shift;   // 128-byte pieces per segment for P1/P2 (1..64, power of two), 0: lane per segment
    uint32_t rec_rows;                   // granules (4 records) reserved per segment: rec_region_dwords(seg_bytes) / 4
    uint32_t* recs;                      // chunk records (qoi_decode_core.h), [block of 64 segments][granule row][lane = segment & 63] x 16 bytes:
                                         // a wavefront's granule row is one contiguous KiB - every record load / store is fully coalesced
    uint32_t* rec_gran;                  // [total_segs + 1] granules (4 records) written per segment
    uint8_t* pixels; size_t pixel_stride;
    // workspace, per global segment q
    ParseRec* parse;           // P1
    uint8_t*  entry_phase;     // S1
    uint32_t* px_off;          // S1
    SlotRec*  slot_rec;        // P2
    uint8_t*  slot_in;         // S2
    uint8_t*  alpha_in;        // S2
    u64*      summary;         // P3  [q][65] symbolic words (slots 0..63, pixel)
    uint32_t* entry;           // S3  [q][65] concrete entry state (table 0..63, pixel)
    uint32_t* fix;             // P4  [q][65] true entry state of q where the check failed
    // per 64-segment group G (two-level chains)
    ParseRec* grp_parse;       // S1 l1
    uint8_t*  grp_phase;       // S1 l2
    uint32_t* grp_off;         // S1 l2
    SlotRec*  grp_slot;        // S2 l1
    uint8_t*  grp_slot_in;     // S2 l2
    uint8_t*  grp_alpha_in;    // S2 l2
    u64*      grp_summary;     // S3 l1 [G][65]
    uint32_t* grp_entry;       // S3 l2 [G][65]
    // run descriptors of the flat images (dec_segments_rec<OCH, true> -> dec_expand_runs)
    uint4*    run_desc;        // [flat_segs][desc_cap] (start pixel, pixels, pixel value, -) of the long runs of a segment, in stream order
    uint32_t* run_cnt;         // [total_segs + 1] descriptors the segment wrote this round (valid for the segments in run_queue)
    uint32_t* run_queue;       // [total_segs] segments that wrote descriptors this round
    uint32_t* run_queue_n;     // [1] their number (word 3 of the counter header; dec_fill zeroes it again)
    uint32_t  desc_cap;        // descriptors a segment can hold: every second record at most ends a run
    uint32_t  flat_segs;       // segments of flat images in this call (0: one launch of dec_segments_rec)
    uint32_t  desc_all;        // 1: the other images leave a descriptor per long QOI_OP_RUN too (in their segments' summary slots)
    uint32_t* first_bad;       // [n_images] min failing segment, 0xFFFFFFFF: none
    uint32_t* pending;         // [1] images that need another round
    uint32_t* redo_segs;       // [1] statistics
    uint32_t refine_inner;     // refinement rounds: repetitions of P3 + S3 before P4 (env QOIMI_DEC_INNER)
    uint32_t first_inner;      // first round: refinement passes (P3 from the speculated entry states + S3) appended for flat images (env QOIMI_DEC_INNER1)
    uint32_t only_flat;        // set by the launcher for those passes: dec_summarize_rec<true> serves flat images only
    // Refinement passes stop by themselves (round 6): the state chain that follows pass s counts the segments whose entry state changed in
    // what the next pass would read from it (the alpha bytes, the slot of the entry pixel) in conv[s]; the kernels of pass s + 1 return at
    // once where that count is zero - the passes behind a fixed point are launches of nothing.  conv: 16 words of the zeroed counter header.
    uint32_t* conv;
    uint32_t  conv_pass;       // set by the launcher: 1 + the pass's number (1..16); 0 = not a refinement pass (nothing counted, nothing skipped)
    // calls of a few large images: the per-image level of the state chain (S3 l2) runs as l2_wgs workgroups per image
    uint32_t  l2_wgs;          // 1: dec_chain_state_l2; 8: dec_chain_state_l2m (ticket / flag words live in the counter header)
    uint32_t  l2_tag_base;     // first tag of this round's S3 launches (a launch's flags carry its tag: nothing to reset)
    u64*      l2_sum;          // [n_images * l2_wgs][65] symbolic summary of a workgroup's groups
    uint32_t* l2_ticket;       // [n_images] workgroups take their place in the image in start order
    uint32_t* l2_flag;         // [n_images * l2_wgs] tag of the launch whose summary stands in l2_sum
    uint32_t p3_plain;         // 1: dec_summarize_rec starts in its one-dword plain form (QOIMI_P3_PLAIN=0 turns it off: diagnostics)
    uint32_t sync_all;         // 1: no look-back synchronisation - every segment takes the full parse (segment sizes the 128-byte piece parse does not cover)
    uint32_t* sync_fails;      // [1] segments whose look-back synchronisation failed in dec_transcode (they take the full parse)
    uint8_t*  sync_fail;       // [total_segs + 1] 1: the segment's entry position is not known yet (dec_transcode<0>)
    // calls of a few images (round 6): pixel offsets and speculated slot / alpha of every segment by ONE single-pass look-back kernel
    // (dec_scan_entry) instead of dec_parse_fine + S1 x 3 + dec_transcode<1> + dec_slot_tails + S2 x 3.  Every image's seg_base is a multiple
    // of kScanSegs then (padding segments in between belong to no image), so a workgroup's segments lie in one image.
    uint32_t  fused;           // 1: that path (the launcher's choice, qoimi_decode_batch)
    uint32_t  epoch;           // the context's number of this call (16 bits are compared): the tag of the look-back words below
    u64*      scan_status;     // [total_segs / kScanSegs] one tagged 8-byte word per workgroup of dec_scan_entry (aggregate / inclusive prefix);
                               // lives in an arena of its own that is zeroed when it is allocated and when the tag wraps, never per call
    uint32_t* scan_ticket;     // [1] workgroups take their place in start order (word 5 of the zeroed counter header)
    // ... and the round's counters straight into pinned HOST words from dec_fill (which then prepares every image's restart in its first
    // wavefront): no copy back behind the round, the host waits for the stream and reads them
    u64*      qtr_summary;     // [total_grps][4][65] calls of a few images: symbolic summary of every quarter (16 segments) of a group - the group levels of the
                               // state chain run as four wavefronts per group (dec_chain_state_l1q / _l3q); nullptr: one wavefront per group
    // ... and the image table in the KERNEL ARGUMENTS of dec_transcode<0> (at most four images), which writes the device copy for the kernels
    // behind it: no host-to-device copy in front of the call (its API call and its blit kernel were 8 of a 4K frame's 182 us).  The counter
    // header is then zeroed by the dec_fill of the context's previous call (qoimi_decode_batch keeps track).
    uint32_t  tab_in_args;     // 1: tab4 holds the table
    DecImage  tab4[4];
    u64*      grp_prefix;      // [total_grps][65] ... and the per-image level without a chain of workgroups (dec_chain_state_l2p): inside every share of groups the
    u64*      share_prefix;    // [n_images * 8 * 16][65] symbolic INCLUSIVE prefixes, likewise over the 16 shares of a workgroup; dec_chain_state_l3q applies
                               // "workgroups in front, shares in front, groups in front" to the image's start state itself: a dozen steps instead of a wait
    // ... and dec_transcode<0> with TWO lanes per segment (128-byte segments only): a lone frame's wavefronts are alone on their SIMDs,
    // each one instruction every ~8 cycles; twice the wavefronts of 60 % the length (32 bytes of run-up + 64 of walk instead of 32 + 128)
    // fill the issue slots.  The halves' records lie in the segment's rows [0, n0) and [tr_rows_half, tr_rows_half + n1);
    // rec_gran = n0 | n1 << 16 (a segment dec_transcode<1> wrote: n1 = 0) - RecSource maps a granule number to its row.
    uint32_t  tr_split;        // 1: that form
    uint32_t  tr_rows_half;    // rows reserved per half
    uint32_t  tr_scan;         // 1: dec_scan_entry's work rides on that launch too (transcode_scan_tail); the kernels behind it return where sync_fails != 0
    uint32_t* s3_ctr;          // [n_images * 8 * 17] arrival counters of dec_chain_state_l1q (round 6): the LAST group of a share to arrive composes the share's
                               // prefixes, the last share of a workgroup's sixteen the share prefixes - the per-image level rides on the group level's
                               // launch (no dec_chain_state_l2p launch); zeroed by the dec_summarize_rec launch in front of every chain; nullptr: l2p
    u64*      share_sum;       // [n_images * 8 * 16][65] the shares' summaries (written through, read by the last arriver)
    uint32_t  tail_fused;      // 1: that form
    uint32_t* host_result;     // pinned host words: [0] pending, [1] redo_segs, [2] sync_fails, [3] 1: some image is still being filled, [4] the call's number (written last)
};
constexpr uint32_t kScanSegs = 256;      // segments per workgroup of dec_scan_entry

void launch_decode_parse(const DecParams& p, hipStream_t st, KernelTimer* tm);
void launch_decode_round(const DecParams& p, int out_channels, bool refine, hipStream_t st, KernelTimer* tm);
void launch_decode_fused_front(const DecParams& p, hipStream_t st, KernelTimer* tm);      // dec_transcode<0> + dec_scan_entry
void launch_decode_parse_rest(const DecParams& p, hipStream_t st, KernelTimer* tm);       // what launch_decode_parse runs behind dec_transcode<0>
void launch_decode_fill(const DecParams& p, int out_channels, hipStream_t st, KernelTimer* tm);
void launch_decode_sequential(const DecParams& p, int out_channels, hipStream_t st, KernelTimer* tm);
constexpr int kMaxSpecRounds = 24;   // speculation rounds before the images still open are finished sequentially

// ---- synthetic frames --------------------------------------------------------------
struct SynthParams {
    uint8_t* pixels; size_t pixel_stride;
    uint32_t npx, width, n_frames, first_frame, seed;
    int kind;
};
void launch_synth(const SynthParams& p, hipStream_t st);
void launch_hash_streams(const uint8_t* streams, size_t stride, const int* lens, uint32_t n, u64* out, hipStream_t st);

}  // namespace qoimi
