// qoi_host.hip — C-ABI shim of libqoi_mi355x.so (include/qoi_mi355x.h).
//
// Part 1 mirrors the reference's public functions (qoi.h:252,265,278,289): identical
// argument validation, malloc()-owned results, NULL / 0 on failure.  Part 2 is the
// additive device-resident batch API.  There is NO CPU codec in this library: every
// pixel/stream byte is produced by the gfx950 kernels, and all entry points fail when
// no GPU is usable.
#include <hip/hip_runtime.h>
#include <sys/mman.h>

#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

// The library is built with -fvisibility=hidden: what the header declares is the whole exported surface (tests/test_abi.py
// reads it back with nm -D).
#pragma GCC visibility push(default)
#include "../../include/qoi_mi355x.h"
#pragma GCC visibility pop
#include "qoi_decode_core.h"
#include "qoi_kernels.h"

using namespace qoimi;

// ------------------------------------------------------------------------------------
// errors
// ------------------------------------------------------------------------------------
static thread_local std::string t_error;
static int fail(int code, const std::string& msg) { t_error = msg; return code; }
#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(e_ == hipErrorOutOfMemory ? QOIMI_E_NOMEM                             \
                        : (e_ == hipErrorNoDevice || e_ == hipErrorInvalidDevice || e_ == hipErrorInsufficientDriver) ? QOIMI_E_NO_GPU \
                        : QOIMI_E_INTERNAL,                                                    \
                        std::string(#expr) + ": " + hipGetErrorString(e_));                  \
    } while (0)

extern "C" const char* qoimi_last_error(void) { return t_error.c_str(); }

// Every entry point works on its context's device and leaves the calling thread's current device as it found it
// (a caller may hold several GPUs, e.g. under torch).
struct DeviceGuard {
    int prev = -1; bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
        else if (prev < 0) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};
extern "C" const char* qoimi_version(void) { return "qoi_mi355x 0.1 gfx950"; }

// ------------------------------------------------------------------------------------
// context: device + growable workspace arenas
// ------------------------------------------------------------------------------------
struct Arena {
    void* base = nullptr;
    size_t cap = 0;
    unsigned gen = 0;              // allocations so far (what a caller that remembers "I zeroed this part" compares)
    int reserve(size_t bytes) {
        if (bytes <= cap) return QOIMI_OK;
        if (base) { (void)hipFree(base); base = nullptr; cap = 0; }
        // (a quarter more than asked for, so that calls of slowly growing batches do not reallocate every time - but no more than 256 MiB:
        // the decode arena of the 1024-frame 4K shard is 45 GB, its margin was another 11)
        const size_t slack = bytes / 4 < ((size_t)256 << 20) ? bytes / 4 : ((size_t)256 << 20);
        size_t want = bytes + slack + (1u << 20);
        HIP_TRY(hipMalloc(&base, want));
        cap = want; ++gen;
        return QOIMI_OK;
    }
    void release() { if (base) (void)hipFree(base); base = nullptr; cap = 0; }
};

struct Carver {   // hands out 256-byte aligned pieces of an arena
    uint8_t* base; size_t off = 0;
    explicit Carver(void* b) : base((uint8_t*)b) {}
    template <class T> T* take(size_t count) {
        off = (off + 255u) & ~(size_t)255u;
        T* p = base ? (T*)(base + off) : nullptr;
        off += count * sizeof(T);
        return p;
    }
};

struct qoimi_ctx {
    int device = 0;
    Arena enc_ws, dec_ws;       // kernel workspaces
    Arena dec_scan;             // look-back words of dec_scan_entry (calls of a few images): tagged with dec_epoch, zeroed when allocated / when the tag wraps
    uint32_t dec_epoch = 0;     // number of the last such call (16 bits are compared)
    struct { void* at = nullptr; unsigned gen = 0; bool valid = false; } dec_hdr_zero;
    void* dec_tail_stream = nullptr; bool dec_tail_open = false;   // a decode call returned on its pinned result words while its last launch was still retiring on this stream   // the counter header the last decode call's dec_fill left zeroed (arena base + generation)
    Arena io_a, io_b, io_c;     // staging for the host-pointer (drop-in) path
    uint32_t* host_word = nullptr;   // pinned words for read-backs
    hipStream_t own_stream = nullptr; // private non-blocking stream: self-test at creation, the drop-in entry points' work
    void* pin_buf = nullptr; size_t pin_cap = 0;   // pinned staging for small host->device tables
    long long dec_stats[4] = {0, 0, 0, 0};
    uint32_t seg_bytes = 0;     // decode segment size; 0: chosen per call from the batch's stream bytes
    uint32_t* last_enc_err = nullptr;   // device flag of the most recent encode launch
    uint32_t* last_enc_err2 = nullptr;  // ... of the other channel group of a qoimi_encode_images call that held 3- and 4-channel images
    void* enc_pin_buf = nullptr; size_t enc_pin_cap = 0; hipEvent_t enc_pin_ev = nullptr;   // pinned staging of qoimi_encode_images' tables (its own: the call is
                                        // asynchronous, decode calls reuse pin_buf at once) and the event behind the last copies out of it
    bool xchg_ordered = false;          // result of the LDS exchange-order self-test (enc_slabs PROBE 1)
    long long enc_calls = 0;            // encode calls so far: the self-test is repeated every enc_recheck_every of them
    long long enc_calls_at_check = 0;   // ... as of the launch of the repeat in flight (or of the last one)
    long long enc_calls_last_passed = 0;   // ... as of the launch of the last repeat that PASSED (0: the test at creation)
    long long enc_recheck_every = 256, enc_suspect_calls = 0;   // env QOIMI_ENC_RECHECK_EVERY
    bool recheck_pending = false;       // a repeated self-test is in flight on own_stream, result in host_word[8]
    bool recheck_failed_unreported = false;   // a repeat failed: the next qoimi_encode_status reports it (once)
    bool test_force_recheck_fail = false;     // env QOIMI_TEST_FORCE_RECHECK_FAIL (tests): every repeat counts as failed
    int enc_ticket = 1, enc_set_slabs = 0, enc_warm = 1;   // tuning / test knobs (env QOIMI_ENC_*)
    bool dropin = false;                // the context of a thread's qoi_encode / qoi_decode calls (thread_ctx)
    int enc_tree_ticket = -1;           // -1: 1 for qoimi_encode_batch, 0 inside the drop-in qoi_encode.  1: tree placement hands its units out by one ticket per workgroup (start order: no assumption about the dispatcher); 0 (QOIMI_ENC_TREE_TICKET=0,
                                        // and always inside the drop-in qoi_encode, which encodes again by itself): by workgroup index, 4 us less per 4K frame
    int dec_tr_scan = 0;                // env QOIMI_DEC_TR_SCAN=1 (experiment, measured SLOWER: 46.6 us against 24.5 + 20.3 on a lone 4K frame, profiles/r06_s15): dec_scan_entry's
                                        // work as the epilogue of the two-lane transcoder instead of a launch of its own
    bool dec_few_longruns = false;      // the context's last call of up to four images on the single-pass path met 1024 long QOI_OP_RUNs or more: the next one takes run descriptors
    bool dec_few_syncfail = false;      // the context's last call of up to four images held segments its transcoder could not synchronise: see decode_some
    int dec_fused_adapt = 1;            // env QOIMI_DEC_FUSED_ADAPT=0: such calls try the single-pass path every time
    bool dec_nonflat_repair = false;    // the context's last call of more than four images (flat ones aside) needed a repair round: see choose_seg_bytes
    int dec_class_split = 1;            // env QOIMI_DEC_CLASS_SPLIT=0: a call that mixes flat images with others is one pass over all of them (round 5)
    int dec_small_seg = 1;              // env QOIMI_DEC_SMALL_SEG=0: calls of a few images never below 128-byte segments
    int dec_conv = 1;                   // env QOIMI_DEC_CONV=0: refinement passes run to their count (1: they stop at a fixed point, DecParams::conv)
    int dec_s3_ride = 0;                // env QOIMI_DEC_S3_RIDE=1 (experiment, measured: 21.6 -> 20.7 us for the two levels on a lone 4K frame, profiles/r06_s14): the per-image
                                        // level of the state chain rides on the group level's launch (last arrivers) instead of dec_chain_state_l2p's own launch
    int dec_split_max = 512;            // env QOIMI_DEC_SPLIT_MAX: the largest segment of a call of a few images that takes two transcoder lanes (128 / 256 / 512 / 1024: a 5120 x 2880 photograph 211 / 200 / 199 / 198 us, a 4K noise frame 208 / 208 / 197 / 199, 8192^2 510 / 519 / 544 / 546 - it takes 1 KiB - profiles/r06_s44_split_max.txt)
    int dec_split = 1;                  // env QOIMI_DEC_SPLIT=0: one transcoder lane per segment in those calls too
    int dec_fused = 1;                  // env QOIMI_DEC_FUSED=0: calls of a few images take the three-level chains of the batch path instead of the single-pass look-back kernels
    uint32_t test_spin_bound = 0;       // env QOIMI_TEST_SPIN_BOUND (tests): polls before a placement wait gives up
    bool tight_buffer = false;          // env QOIMI_ENCODE_TIGHT_BUFFER=1 (read once, at creation): qoi_encode sizes its result by the thread's previous stream instead of
                                        // returning the reference's worst-case allocation (qoi.h:374-379)
    int enc_gen_slabs = 0;              // env QOIMI_ENC_GEN_SLABS (1..16): slabs per set of the pass over flagged images; 0: kEncGenSetSlabs, twice that for
                                        // calls of 3 x 65536 slabs and more (8 / 12 / 16 slabs, 1024 frames: constant 7.69 / 7.34 / 6.75 ms, uiflat 20.62 / 20.49 / 20.34,
                                        // 512 sprites 8.86 / 8.70 / 8.76 - profiles/r05_s22_enc_gen_slabs16.txt; a single frame has too few sets for that)
    int enc_gen_small_div = 0;          // env QOIMI_ENC_GEN_GRID_DIV (0: 32)
    int enc_gen_grid_div = 1;           // (32 / 4 / 1: uiflat 21.3 / 21.2 / 20.4 ms, sprite_alpha 11.0 / 11.1 / 10.1 per 512, profiles/r05_s14_enc_grid.txt) env QOIMI_ENC_GEN_GRID_HOT: the pass over flagged images runs with 1/N of its units when the previous batch held flagged images
    int enc_uni = -1;                   // one encode pass, sets whose look-back window does not do take the state look-back one by one.  -1: for calls of a few
                                        // images (tree placement) behind a call that met flat stretches (host_word[14]); env QOIMI_ENC_UNI=1 always / 0 never
    int enc_prezero = 1;                // env QOIMI_ENC_PREZERO=0: calls of a few images zero their records with hipMemsetAsync every time (see enc_sets: zero_next)
    struct { void* ptr = nullptr; size_t bytes = 0; unsigned gen = 0; long long seq = -1; bool valid = false; } prezero;   // the region the last such call zeroed for its successor
    long long enc_ws_seq = 0;           // calls that laid out the encode workspace so far (a zeroed region is good for the very next one only)
    int enc_parity = 0;                 // which of the two regions the next call of a few images takes
    int enc_all_g2 = 1;                 // env QOIMI_ENC_ALL_G2=0: a batch behind a batch of flagged images only still runs its first pass (with a sixteenth of its workgroups)
    int enc_g2 = 1;                     // env QOIMI_ENC_G2=0: flagged images (flat content) go through the summary passes instead of the state look-back (ENTRY 2)
    uint32_t enc_epoch = 0;             // encode call number: the tag of the state look-back's granules
    void* g2_zeroed_at = nullptr; size_t g2_zeroed_bytes = 0; unsigned g2_zeroed_gen = 0;     // where those granules were last zeroed
    int enc_adapt = 1;                  // env QOIMI_ENC_ADAPT=0: the set size ignores what the previous call's streams looked like
    uint32_t enc_hint_images = 0;       // images of the batch call whose count of flagged images stands in host_word[13]
    uint32_t enc_hint_npx = 0;          // pixels per image of the batch call whose first stream length stands in host_word[12] (0: none)
    bool enc_heavy_before = false, enc_flagged_before = false;   // what the batch BEFORE the previous one looked like: a hint acts only when two batches in a row agree
    struct { const void* px; size_t ps; qoi_desc desc; int n; void* out; size_t os; int* len; void* st; bool valid = false; } last_enc;   // the last qoimi_encode_batch (qoimi_encode_status re-encodes it order-free if a wait gave up)
    int enc_spread = 1;                 // env QOIMI_ENC_SPREAD: the wavefronts of a workgroup take their tickets from consecutive images (0: all four from one image)
    int enc_pipe = 0;                   // env QOIMI_ENC_PIPE=1 (experiment, with QOIMI_ENC_PERSIST): next set's loads ahead of the current set's placement
    int enc_persist = 0;                // env QOIMI_ENC_PERSIST: workgroups of the first encode pass (0: one per unit)
    int enc_lookback = -1;              // 1: sets place their bytes themselves (decoupled look-back); 2: the same by the tree of byte counts; 0: order-free (scratch slots + enc_offsets + enc_compact); -1: by the call's shape
    std::string enc_debug_dump;         // env QOIMI_ENC_DEBUG_DUMP: file that receives the entry-state arrays of every encode call
    std::string dec_debug_dump;         // env QOIMI_DEC_DEBUG_DUMP: file that receives the per-segment arrays (granule counts, parse records, pixel offsets) of every decode call
    int dec_refine = 1;                 // 0: rounds after a failed check re-speculate from scratch (no alpha hints)
    int dec_fine = 1;                   // 0: lane-per-segment P1/P2 even where the 128-byte piece kernels apply
    int dec_p3_plain = 1, dec_inner = 8, dec_inner1 = 3;   // env QOIMI_P3_PLAIN, QOIMI_DEC_INNER, QOIMI_DEC_INNER1 (read once, at creation)
                                                           // (dec_inner 4 / 8 / 16 on 1024 UI frames: 3 / 2 / 2 rounds in 19.4 / 18.6 / 23.2 ms, profiles/r05_s9_dec_uiflat_inner.txt)
    int dec_l2_wgs = 1;          // dec_chain_state_l2m: 0 never, 1 for calls of up to four images of 128 groups or more, 2 for every call of up to four images (env QOIMI_DEC_L2M, tests)
    int dec_flat_seg = 1;        // 0: calls of flat images take the segment size of the general cost model (env QOIMI_DEC_FLAT_SEG, A/B)
    int dec_run_desc = 2;        // env QOIMI_DEC_RUN_DESC - 0: every long run is written lane by lane; 1: run descriptors for flat images; 2: and a descriptor per long QOI_OP_RUN chunk of the other images
    int dec_max_rounds = kMaxSpecRounds;   // speculation rounds before the sequential last resort (env QOIMI_DEC_MAX_ROUNDS, tests)
    size_t last_drop_len = 0;           // length of the last stream the drop-in qoi_encode returned on this context (page populate-ahead)
    long long enc_retries = 0;          // calls qoimi_encode_status encoded again order-free after a placement wait gave up
    long long dec_seq_images = 0;       // images finished by dec_sequential since the context was created
    size_t dec_rec_cap = (size_t)16 << 30;   // largest record arena: a call whose streams need more is decoded in sub-batches (set from the device's memory at creation)
    KernelTimer timer;                  // optional per-kernel HIP-event timing
    double prof_ms[kT_count] = {0};     // accumulated kernel milliseconds since profiling was (re)enabled
    long long prof_calls[kT_count] = {0};
};

static const size_t kPixelCap = 400000000u;   // QOI_PIXELS_MAX, qoi.h:332

static bool desc_ok(const qoi_desc* d) {       // qoi.h:366-369 / 514-518
    return d && d->width != 0 && d->height != 0 && d->channels >= 3 && d->channels <= 4 &&
           d->colorspace <= 1 && d->height < kPixelCap / d->width;
}

extern "C" size_t qoimi_encode_bound(const qoi_desc* desc) {
    if (!desc_ok(desc)) return 0;
    return (size_t)desc->width * desc->height * (desc->channels + 1u) + kHeaderBytes + kTrailerBytes;
}

extern "C" int qoimi_ctx_create(int device, qoimi_ctx** out) {
    if (!out) return fail(QOIMI_E_ARG, "out is NULL");
    *out = nullptr;
    int n = 0;
    HIP_TRY(hipGetDeviceCount(&n));
    if (device < 0 || device >= n) return fail(QOIMI_E_NO_GPU, "no such GPU device");
    DeviceGuard guard(device);
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(QOIMI_E_NO_GPU, std::string("kernels are built for gfx950 only, device is ") + prop.gcnArchName);
    qoimi_ctx* c = new qoimi_ctx();
    c->device = device;
    {   // record arena of a decode call: up to a sixth of the device's memory (48 GiB on a 288 GB MI355X: the 1024-frame shard of
        // BASELINE configs[4] in one piece, 45.1 GB of workspace = 4.3 x its stream bytes at 36.4 ms), never less than 1 GiB.  A caller
        // short of device memory caps it (QOIMI_DEC_REC_CAP_MB): 24 GiB = two sub-batches, 27.4 GB = 2.6 x the stream bytes at 36.8 ms
        // (+1 %: every kernel's tail twice), 16 GiB = three, 18.3 GB at 37.1 ms (profiles/r05_s17_dec_cap.txt, r05_s27_dec_cap_wall.txt).
        // The default was 24 GiB for sessions 24-34 of round 5: 172.8 Gpx/s (median of six runs on five boxes) against 176 in one piece.
        size_t cap = (size_t)prop.totalGlobalMem / 6u;
        if (cap > ((size_t)48 << 30)) cap = (size_t)48 << 30;
        if (cap < ((size_t)1 << 30)) cap = (size_t)1 << 30;
        c->dec_rec_cap = cap;
    }
    if (hipHostMalloc((void**)&c->host_word, 256) != hipSuccess) { delete c; return fail(QOIMI_E_NOMEM, "hipHostMalloc failed"); }
    // Measure (do not assume) the LDS conflict order the fast colour-table probe relies on.
    {
        hipStream_t pst = nullptr;
        if (hipStreamCreateWithFlags(&pst, hipStreamNonBlocking) != hipSuccess) { (void)hipHostFree(c->host_word); delete c; return fail(QOIMI_E_NO_GPU, "hipStreamCreate failed"); }
        c->own_stream = pst;           // also the stream of the drop-in entry points (one context per calling thread)
        c->xchg_ordered = run_lds_order_selftest(pst) == 0;
    }
    c->host_word[12] = 0u; c->host_word[13] = 0u; c->host_word[14] = 0u; c->host_word[15] = 0u;
    // Documented settings: the order-independent colour-table probe from the start; the tight result buffer of qoi_encode.
    if (const char* e = getenv("QOIMI_ENC_PROBE")) { if (atoi(e) == 0) c->xchg_ordered = false; }
    if (const char* e = getenv("QOIMI_ENCODE_TIGHT_BUFFER")) c->tight_buffer = atoi(e) != 0;
    // Everything else in the environment is a measurement / test knob and is looked at only under QOIMI_TUNING=1 (tests/conftest.py and
    // tools/measure set it): an inherited QOIMI_* variable cannot change kernels, segment sizes or placement of a production process.
    const char* tune = getenv("QOIMI_TUNING");
    if (tune && atoi(tune) != 0) {
        auto knob = [](const char* name, int& v) { if (const char* e = getenv(name)) v = atoi(e); };
        auto flag = [](const char* name, int& v) { if (const char* e = getenv(name)) v = atoi(e) != 0; };
        if (const char* e = getenv("QOIMI_ENC_RECHECK_EVERY")) { long v = atol(e); if (v >= 1) c->enc_recheck_every = v; }
        knob("QOIMI_ENC_TICKET", c->enc_ticket); knob("QOIMI_ENC_SET_SLABS", c->enc_set_slabs); knob("QOIMI_ENC_WARM", c->enc_warm);
        knob("QOIMI_ENC_LOOKBACK", c->enc_lookback);
        if (const char* e = getenv("QOIMI_ENC_DEBUG_DUMP")) c->enc_debug_dump = e;
        if (const char* e = getenv("QOIMI_DEC_DEBUG_DUMP")) c->dec_debug_dump = e;
        flag("QOIMI_ENC_SPREAD", c->enc_spread); flag("QOIMI_ENC_TREE_TICKET", c->enc_tree_ticket); flag("QOIMI_ENC_ADAPT", c->enc_adapt);
        flag("QOIMI_ENC_G2", c->enc_g2); flag("QOIMI_ENC_ALL_G2", c->enc_all_g2); flag("QOIMI_ENC_PREZERO", c->enc_prezero); flag("QOIMI_ENC_UNI", c->enc_uni);
        flag("QOIMI_ENC_PIPE", c->enc_pipe);
        if (const char* e = getenv("QOIMI_ENC_GEN_GRID_DIV")) { const int v = atoi(e); if (v >= 1) c->enc_gen_small_div = v; }
        if (const char* e = getenv("QOIMI_ENC_GEN_GRID_HOT")) { const int v = atoi(e); if (v >= 1) c->enc_gen_grid_div = v; }
        if (const char* e = getenv("QOIMI_ENC_GEN_SLABS")) { const int v = atoi(e); if (v >= 1 && v <= (int)kEncMaxSetSlabs) c->enc_gen_slabs = v; }
        if (const char* e = getenv("QOIMI_ENC_PERSIST")) { const int v = atoi(e); if (v >= 0) c->enc_persist = v; }
        knob("QOIMI_DEC_FINE", c->dec_fine); knob("QOIMI_DEC_REFINE", c->dec_refine); knob("QOIMI_P3_PLAIN", c->dec_p3_plain);
        if (const char* e = getenv("QOIMI_DEC_INNER")) { const int v = atoi(e); if (v >= 0 && v <= 64) c->dec_inner = v; }
        if (const char* e = getenv("QOIMI_DEC_INNER1")) { const int v = atoi(e); if (v >= 0 && v <= 64) c->dec_inner1 = v; }
        knob("QOIMI_DEC_L2M", c->dec_l2_wgs); knob("QOIMI_DEC_RUN_DESC", c->dec_run_desc); knob("QOIMI_DEC_FLAT_SEG", c->dec_flat_seg);
        knob("QOIMI_DEC_FUSED", c->dec_fused); knob("QOIMI_DEC_SPLIT", c->dec_split); knob("QOIMI_DEC_S3_RIDE", c->dec_s3_ride); knob("QOIMI_DEC_TR_SCAN", c->dec_tr_scan); knob("QOIMI_DEC_CONV", c->dec_conv); knob("QOIMI_DEC_SMALL_SEG", c->dec_small_seg); knob("QOIMI_DEC_CLASS_SPLIT", c->dec_class_split); knob("QOIMI_DEC_FUSED_ADAPT", c->dec_fused_adapt);
        if (const char* e = getenv("QOIMI_DEC_SPLIT_MAX")) { const int v = atoi(e); if (v >= 64 && v <= 4096) c->dec_split_max = v; }
        if (const char* e = getenv("QOIMI_DEC_MAX_ROUNDS")) { int v = atoi(e); if (v >= 1) c->dec_max_rounds = v; }
        if (const char* e = getenv("QOIMI_DEC_REC_CAP_MB")) { long v = atol(e); if (v >= 1) c->dec_rec_cap = (size_t)v << 20; }
        if (const char* e = getenv("QOIMI_SEG_BYTES")) { long v = atol(e); if (v >= 64 && v <= (1 << 20)) c->seg_bytes = (uint32_t)v; }
    }
#ifdef QOIMI_TEST_HOOKS
    // failure injection, compiled into the test flavour of the library only (make TEST_HOOKS=1 -> libqoi_mi355x_test.so)
    if (const char* e = getenv("QOIMI_TEST_FORCE_RECHECK_FAIL")) c->test_force_recheck_fail = atoi(e) != 0;
    if (const char* e = getenv("QOIMI_TEST_SPIN_BOUND")) { const long v = atol(e); if (v >= 1) c->test_spin_bound = (uint32_t)v; }
#endif
    *out = c;
    return QOIMI_OK;
}

extern "C" void qoimi_ctx_destroy(qoimi_ctx* c) {
    if (!c) return;
    DeviceGuard guard(c->device);
    (void)hipDeviceSynchronize();       // calls still in flight write to the arenas and to the pinned words freed below
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    c->enc_ws.release(); c->dec_ws.release(); c->dec_scan.release(); c->io_a.release(); c->io_b.release(); c->io_c.release();
    if (c->host_word) (void)hipHostFree(c->host_word);
    if (c->pin_buf) (void)hipHostFree(c->pin_buf);
    if (c->enc_pin_buf) (void)hipHostFree(c->enc_pin_buf);
    if (c->enc_pin_ev) (void)hipEventDestroy(c->enc_pin_ev);
    delete c;
}

extern "C" int qoimi_set_decode_record_cap(qoimi_ctx* c, size_t bytes, int release) {
    if (!c || bytes < ((size_t)1 << 20)) return fail(QOIMI_E_ARG, "record cap: NULL context or less than 1 MiB");
    c->dec_rec_cap = bytes;
    if (release) {
        DeviceGuard guard(c->device);
        (void)hipDeviceSynchronize();
        c->dec_ws.release();
    }
    return QOIMI_OK;
}

// Per-kernel timing with HIP events on the launch stream.  on=1 resets the accumulators.
extern "C" int qoimi_set_profiling(qoimi_ctx* c, int on) {
    if (!c) return fail(QOIMI_E_ARG, "ctx is NULL");
    DeviceGuard guard(c->device);
    if (on && !c->timer.created) {
        for (int i = 0; i < KernelTimer::kMax; ++i) HIP_TRY(hipEventCreate(&c->timer.ev[i]));
        c->timer.created = true;
    }
    c->timer.on = on != 0;
    c->timer.n = 0;
    if (on) for (int i = 0; i < kT_count; ++i) { c->prof_ms[i] = 0; c->prof_calls[i] = 0; }
    return QOIMI_OK;
}

// fold the recorded events into the accumulators (the stream must be idle)
static void timer_collect(qoimi_ctx* c) {
    {
        KernelTimer& t = c->timer;
        int open_total = -1;                           // index of the kT_begin a kT_enc_total / kT_dec_total mark closes
        for (int i = 0; i < t.n; ++i) {
            if (t.tag[i] == kT_begin) { if (open_total < 0) open_total = i; continue; }
            float ms = 0;
            if (t.tag[i] == kT_enc_total || t.tag[i] == kT_dec_total) {
                if (open_total >= 0 && hipEventElapsedTime(&ms, t.ev[open_total], t.ev[i]) == hipSuccess) { c->prof_ms[t.tag[i]] += ms; c->prof_calls[t.tag[i]] += 1; }
                open_total = -1;
                continue;
            }
            if (i > 0 && hipEventElapsedTime(&ms, t.ev[i - 1], t.ev[i]) == hipSuccess) { c->prof_ms[t.tag[i]] += ms; c->prof_calls[t.tag[i]] += 1; }
        }
        t.n = 0;
    }
}

// Synchronises `stream`, then copies accumulated milliseconds and launch counts per kernel
// (index = position in qoimi_kernel_name).  Returns the number of kernels.
extern "C" int qoimi_get_profile(qoimi_ctx* c, void* stream, double* ms, long long* calls, int cap) {
    if (!c) return fail(QOIMI_E_ARG, "ctx is NULL");
    DeviceGuard guard(c->device);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    timer_collect(c);
    for (int i = 0; i < kT_count && i < cap; ++i) { if (ms) ms[i] = c->prof_ms[i]; if (calls) calls[i] = c->prof_calls[i]; }
    return kT_count;
}

extern "C" const char* qoimi_kernel_name(int i) {
    static const char* names[kT_count] = {"", "enc_slab_summary", "enc_scan_groups", "enc_scan_images", "enc_slabs", "enc_slabs_generic", "enc_offsets", "enc_compact",
        "dec_parse", "dec_chain_parse", "dec_transcode", "dec_chain_slots", "dec_summarize", "dec_chain_state",
        "dec_segments", "dec_prepare_restart", "dec_fill", "dec_expand_runs", "encode_total", "decode_total"};
    return (i >= 0 && i < kT_count) ? names[i] : "";
}

extern "C" long long qoimi_encode_suspect_calls(qoimi_ctx* c) { return c ? c->enc_suspect_calls : 0; }
extern "C" long long qoimi_encode_retries(qoimi_ctx* c) { return c ? c->enc_retries : 0; }
extern "C" int qoimi_set_encode_small_call_order(qoimi_ctx* c, int by_workgroup_index) {
    if (!c) return fail(QOIMI_E_ARG, "ctx is NULL");
    c->enc_tree_ticket = by_workgroup_index ? 0 : 1;
    return QOIMI_OK;
}

// device memory the context holds: [0] encode workspace, [1] decode workspace, [2] staging of the host-pointer entry points
extern "C" void qoimi_workspace_bytes(qoimi_ctx* c, size_t out[3]) {
    out[0] = c ? c->enc_ws.cap : 0; out[1] = c ? c->dec_ws.cap : 0;
    out[2] = c ? c->io_a.cap + c->io_b.cap + c->io_c.cap : 0;
}

extern "C" void qoimi_decode_stats(qoimi_ctx* c, long long out[4]) {
    for (int i = 0; i < 4; ++i) out[i] = c ? c->dec_stats[i] : 0;
}

// ------------------------------------------------------------------------------------
// encode
// ------------------------------------------------------------------------------------
extern "C" int qoimi_encode_batch(qoimi_ctx* c, const void* d_pixels, size_t pixel_stride,
                                  const qoi_desc* desc, int n_images,
                                  void* d_streams, size_t stream_stride, int* d_stream_len,
                                  void* stream) {
    if (!c || !d_pixels || !d_streams || !d_stream_len || n_images <= 0) return fail(QOIMI_E_ARG, "NULL/empty argument");
    if (!desc_ok(desc)) return fail(QOIMI_E_ARG, "descriptor rejected (qoi.h:364-372 rules)");
    const size_t npx = (size_t)desc->width * desc->height;
    if (pixel_stride < npx * desc->channels) return fail(QOIMI_E_ARG, "pixel_stride smaller than one image");
    if (stream_stride < qoimi_encode_bound(desc)) return fail(QOIMI_E_ARG, "stream_stride smaller than qoimi_encode_bound");
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)stream;

    EncParams p;
    memset(&p, 0, sizeof p);
    p.pixels = (const uint8_t*)d_pixels; p.pixel_stride = pixel_stride;
    p.npx = (uint32_t)npx; p.n_images = (uint32_t)n_images;
    p.spi = (uint32_t)((npx + kEncSlabPx - 1) / kEncSlabPx);
    p.gpi = (p.spi + 63u) / 64u;
    p.width = desc->width; p.height = desc->height; p.channels = desc->channels; p.colorspace = desc->colorspace;
    // The ordered-exchange probe rests on a measured hardware property (qoi_encode.hip): measure it again as the context
    // lives on.  The repeat runs on the context's private stream; its result is looked at by the next call.
    if (c->recheck_pending && hipStreamQuery(c->own_stream) == hipSuccess) {
        c->recheck_pending = false;
        if (c->host_word[8] != 0u || c->test_force_recheck_fail) {
            // Never observed.  The context switches to the order-free probe for good and THIS call is encoded with it (its own
            // stream is sound, the call does not fail); what cannot be undone is reported: every call since the launch of the last
            // repeat that PASSED is suspect - the ones before the failed repeat was launched and the ones made while it ran
            // (enc_calls still excludes the call at hand) - and the next qoimi_encode_status returns QOIMI_E_INTERNAL once.
            c->xchg_ordered = false;
            c->enc_suspect_calls += c->enc_calls - c->enc_calls_last_passed;
            c->enc_calls_last_passed = c->enc_calls;
            c->recheck_failed_unreported = true;
            (void)fail(QOIMI_E_INTERNAL, "the LDS exchange-order self-test failed on repetition: streams encoded since the last passed check are suspect (qoimi_encode_suspect_calls); this context now uses the order-free probe");
        } else {
            c->enc_calls_last_passed = c->enc_calls_at_check;
        }
    }
    ++c->enc_calls;
    if (c->xchg_ordered && !c->recheck_pending && c->enc_calls - c->enc_calls_at_check >= c->enc_recheck_every && c->io_c.reserve(256) == QOIMI_OK) {
        c->enc_calls_at_check = c->enc_calls;
        uint32_t* d_flag = (uint32_t*)c->io_c.base + 32;
        launch_lds_order_selftest(d_flag, c->own_stream);
        if (hipMemcpyAsync(&c->host_word[8], d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->own_stream) == hipSuccess) c->recheck_pending = true;
    }
    p.probe_xchg = c->xchg_ordered ? 1 : 0;
    p.use_ticket = c->enc_ticket ? 1 : 0;
    p.warm = c->enc_warm ? 1 : 0;
    p.persist = (uint32_t)c->enc_persist;
    bool all_flagged_before = false;
    p.pipe = (uint32_t)c->enc_pipe;
    p.spread = (uint32_t)c->enc_spread;
    // Slabs per set and placement - functions of the call's shape only; QOIMI_ENC_SET_SLABS / QOIMI_ENC_LOOKBACK force them; every
    // combination gives the same bytes.
    // A wavefront carries the colour table and its staged bytes from slab to slab, so the entry-state replay and the placement are
    // paid once per set - as long as the sets still fill the 256 CUs x 24 wavefronts several times.
    // Look-back (1): a set finds its place in the stream by decoupled look-back over the earlier sets of its image and writes its
    // bytes once, straight from the LDS (sets of more than ~1.4 bytes per pixel spill to a scratch slot and move that part
    // themselves).  The inclusive prefixes travel 64 sets per poll (~1 us) through the sets of an image that finish at about the same
    // time - the ~6000 resident wavefronts divided by the number of images.  With a batch that is a few sets; with ONE image it is
    // all of them (a 4K frame: 71-131 us against 45 order-free), and its ticket counter serves every wavefront in turn.
    // Tree (2, round 4; encode_set): calls of fewer than 8 images.  Every set adds three windows of byte counts - its group's, its
    // block's group totals, the image's block totals - nothing travels from set to set, the sets go by workgroup index.  One frame,
    // tree against the best of the other two forms (profiles/r04_s12_single_placement.txt): 640 x 360 20.2 us / 22.8, 1280 x 720
    // 21.9 / 26.8, 1920 x 1080 27.2 / 27.6, 2560 x 1440 32.0 / 31.4, 3840 x 2160 39.9 / 42.6, 5120 x 2880 50.3 / 57.0.
    // Order-free (0): every set parks its bytes in a scratch slot, enc_offsets scans the sizes with a whole workgroup, enc_compact
    // places them (two more launches, a round trip through scratch).  No set ever waits: what very large images take (16384 x
    // 16384: 552 us against 653 by the tree - 6000 sets in flight, each a few microseconds in its slot waiting for the totals).
    {
        const size_t total_slabs = (size_t)n_images * p.spi;
        uint32_t r = total_slabs >= 3u * 65536u ? 3u : (total_slabs >= 16384u ? 2u : 1u);
        int place = c->enc_lookback >= 0 ? (c->enc_lookback > 2 ? 1 : c->enc_lookback) : (n_images >= 8 ? 1 : 2);
        if (place == 2 && c->enc_lookback < 0) {
            const uint32_t rt = total_slabs < 1500u ? 1u : (total_slabs < 6000u ? 2u : 3u);       // measured above
            if ((p.spi + rt - 1u) / rt > kEncTreeMaxSets) place = 0; else r = rt;
        }
        // Look-back batches: three slabs per set is the size for ~1.2 bytes per pixel - above ~1.4 a set outgrows its 6.3 KB staging
        // buffer and sends what it has through a scratch slot (a second trip through memory for those bytes).  Two slabs stay staged
        // up to 3 bytes per pixel (photo_hard, 2.1 B/px, 128 frames: 3.00 ms at three slabs, 2.77 at two, 3.86 at one;
        // photographs of 1.2 B/px lose 10 % at two).  What the content looks like is taken from the previous batch call of the context:
        // the length of its first stream, copied to a pinned word behind that call (read here without a wait: a stale or missing
        // value only picks the other set size, the streams are the same bytes either way).
        // (round 6: only when the TWO batches before this one were both that heavy.  Photographs behind a batch of 2.1 B/px lost 23 % to the
        // two-slab sets their predecessor had earned, bench.py "alternating", profiles/r06_s8; a workload that alternates now never takes a
        // hint, one that stays with its content takes it from its third batch on)
        if (c->enc_adapt && place == 1 && n_images >= 8) {
            const bool heavy = c->enc_hint_npx != 0u && c->host_word[12] != 0u && (double)c->host_word[12] > 1.4 * (double)c->enc_hint_npx;
            if (r == 3u && heavy && c->enc_heavy_before) r = 2u;
            c->enc_heavy_before = heavy;
        }
        if (c->enc_set_slabs > 0) r = (uint32_t)c->enc_set_slabs;
        if (r > kEncMaxSetSlabs) r = kEncMaxSetSlabs;
        p.set_slabs = r;
        p.set_px = r * kEncSlabPx;
        p.sets_per_image = (p.spi + r - 1u) / r;
        p.set_stride = r * kEncSlabWorst + 16u;
        if (place == 2 && p.sets_per_image > 64u * 64u * 64u) place = 0;            // (three levels of 64; the generic pass has fewer sets)
        p.lookback = (uint8_t)place;
        // Tree: units by workgroup index (a wait is for lower-numbered sets, which the dispatcher started earlier - true of one launch
        // on an idle device; two launches from different streams could in principle hold each other's predecessors out: the waits are
        // bounded - 2^15 polls, tens of milliseconds, where a set's predecessors finish within microseconds - a tripped bound ends every
        // wait of the launch and the call is encoded again order-free by qoi_encode / qoimi_encode_status).  That form is what the
        // drop-in qoi_encode takes (it reads the error word and encodes again by itself).  qoimi_encode_batch (round 6: the default)
        // hands the units out by one ticket per workgroup - START order, no assumption about the dispatcher, so a caller that only
        // synchronises its stream never reads a truncated stream - 4 us more per 4K frame (46.8 against 42.6 us, 720p 23.8 against
        // 21.5: profiles/r05_s1_single_ticket.txt).
        if (place == 2) { p.spread = 0; p.use_ticket = (c->enc_tree_ticket >= 0 ? c->enc_tree_ticket != 0 : !c->dropin) ? 1 : 0; }
    }
    const int place = p.lookback;
    const bool lookback = place != 0;
    p.spin_bound = (place == 2 && !p.use_ticket) ? (1u << 15) : (1u << 22);
    if (c->test_spin_bound) p.spin_bound = c->test_spin_bound;            // tests: make a wait give up
    const size_t T = (size_t)p.n_images * p.spi, G = (size_t)p.n_images * p.gpi, S = (size_t)p.n_images * p.sets_per_image;
    if (T > 0xFFFFFFF0ull) return fail(QOIMI_E_ARG, "batch too large (slab index overflows 32 bits)");
    // Scratch.  Order-free: every set parks its bytes in a slot of its own until the placement passes run (few large images:
    // tens of megabytes).  Look-back: only sets that outgrow their LDS staging buffer (more than ~1.5 bytes per pixel) hold scratch,
    // from their first spill to their copy-out - a pool of kEncPoolSlots slots (more than the wavefronts in flight; fewer for calls
    // of fewer sets), handed out on the device (pool_take).  The 1024-frame 4K shard: 0.67 GB (slots of sixteen slabs) instead of 42.5 GB.
    p.pool = lookback ? 1 : 0;
    p.gen_slabs = c->enc_gen_slabs > 0 ? (uint32_t)c->enc_gen_slabs : ((size_t)n_images * p.spi >= 3u * 65536u ? 2u * kEncGenSetSlabs : kEncGenSetSlabs);
    p.gen_grid_div = (c->enc_adapt && n_images >= 8 && c->host_word[13] != 0u) ? (uint32_t)c->enc_gen_grid_div : 0u;
    p.gen_small_div = (uint32_t)c->enc_gen_small_div;
    // The previous batch held flagged images ONLY (flat content: host_word[13] counts them): this call's first pass will most likely find
    // an image's first flat stretch within microseconds and every other set of the image has nothing to do but to see the flag - one
    // workgroup per four sets is 345 000 workgroups that start and end for 512 4K frames, 0.5 ms of dispatch.  A sixteenth of them, each
    // looking at sixteen units, sees the same flags (photographs pay 7-9 % with several sets per wavefront: the hint is gone after one call).
    if (c->enc_adapt && place == 1 && n_images >= 8) {
        const bool flagged = c->enc_hint_images != 0u && c->host_word[13] >= c->enc_hint_images;
        all_flagged_before = flagged && c->enc_flagged_before;                  // (two batches in a row, as the set size above)
        c->enc_flagged_before = flagged;
    }
    if (all_flagged_before && p.persist == 0u) p.persist = 0xFFFFFFFFu;            // resolved below, once the units are known
    // ... or not at all (QOIMI_ENC_ALL_G2, default on): the pass over flagged images takes EVERY image of this call, and counts the images in
    // which some set had to walk the groups in front of its tail - the same statistic, so a batch of photographs behind flat batches
    // runs once through that pass (sixteen-slab sets, every set through the pool) and hands the next batch back to the two passes.
    p.all_g2 = (all_flagged_before && c->enc_all_g2) ? 1u : 0u;
    const size_t S_gen = (size_t)p.n_images * ((p.spi + p.gen_slabs - 1u) / p.gen_slabs);
    if (lookback) {
        size_t slots = (S + 63u) & ~(size_t)63u;
        p.pool_slots = (uint32_t)(slots < kEncPoolSlots ? slots : kEncPoolSlots);
        const uint32_t r_max = p.set_slabs > p.gen_slabs ? p.set_slabs : p.gen_slabs;     // the generic pass draws on the same pool
        p.set_stride = r_max * kEncSlabWorst + 16u;
    }

    size_t g2_bytes = 0;
    // calls of a few images: two regions of records / tickets / flags / pool map, used in turn - the first launch of a call zeroes the region
    // of the next (enc_sets: zero_next), which then skips its hipMemsetAsync if it is the very next user of the workspace and lays out alike
    const long long ws_seq = ++c->enc_ws_seq;
    const auto zeroed_before = c->prezero;
    c->prezero.valid = false;
    const bool pingpong = place == 2 && c->enc_prezero && p.warm && p.probe_xchg;
    uint8_t* zero_other = nullptr; size_t zero_len = 0;
    // one of the last eight small calls the DEVICE has started met flat stretches (host_word[14]: number of the last call that did,
    // [15]: of the last call started; the calls of a pipeline are set up long before their predecessors run - read once: the device may be writing)
    const uint32_t hint = *(volatile uint32_t*)&c->host_word[14], started = *(volatile uint32_t*)&c->host_word[15];
    const bool hot = hint != 0u && (((started - hint) & 0x1FFFFFFFu) < 8u || ((hint - started) & 0x1FFFFFFFu) < 8u);
    for (int pass = 0; pass < 2; ++pass) {      // pass 0 measures, pass 1 carves
        Carver w(pass ? c->enc_ws.base : nullptr);
        p.status = w.take<u64>(S); p.ticket = w.take<uint32_t>((size_t)n_images); p.err = w.take<uint32_t>(1);
        p.need_generic = w.take<uint32_t>((size_t)n_images); p.any_generic = w.take<uint32_t>(1);
        p.status_gen = w.take<u64>(lookback ? S_gen : 0); p.ticket_gen = w.take<uint32_t>(lookback ? (size_t)n_images : 0);
        {   // tree placement: totals of the groups of 64 sets and of the blocks of 64 groups, for the first pass and for the generic one
            const size_t n1 = (p.sets_per_image + 63u) / 64u, n2 = (n1 + 63u) / 64u;
            const size_t sg1 = (S_gen / (size_t)n_images + 63u) / 64u, sg2 = (sg1 + 63u) / 64u;
            const size_t on = place == 2 ? (size_t)n_images : 0;
            p.tree1 = w.take<u64>(on * n1); p.tree2 = w.take<u64>(on * n2);
            p.tree1_gen = w.take<u64>(on * sg1); p.tree2_gen = w.take<u64>(on * sg2);
        }
        p.pool_map = w.take<u64>(lookback ? (size_t)(p.pool_slots / 64u) * kEncPoolMapStride : 0);
        const size_t zero_bytes = w.off;
        // flagged images (flat content): by state look-back over their sets (ENTRY 2: 520 bytes per set of eight slabs, tagged with the
        // call's number instead of being zeroed) - or, for order-free calls and the order-independent probe, by the summary passes
        // (per slab 2 x (256 B table + 8 B valid + 4 B position): 4.3 GB for the 1024-frame 4K shard)
        const bool g2 = lookback && p.probe_xchg && p.warm && c->enc_g2;
        // One pass or two: a frame of photographic content is 2 us faster through the two-pass kernel, which never needs its second pass
        // (35.3 against 37.1 us per 4K frame); a frame with flat stretches saves the second launch and the first pass's wasted walk with
        // the one-pass kernel (4K: constant 110 -> 93 us, UI 156 -> 111, soft-alpha sprite 126 -> 69; profiles/r05_s16_single_uni.txt).
        // Calls of a few images take the one pass when one of the last eight such calls the device has started met a flat stretch: its first
        // such set left the call's number in a pinned word (leave_hint).  Batches keep two passes (1024 photographs 13.4 against 12.3 ms in one pass).
        p.uni = (g2 && (c->enc_uni > 0 || (c->enc_uni < 0 && c->enc_adapt && place == 2 && hot))) ? 1u : 0u;
        p.host_hint = place == 2 ? &c->host_word[14] : nullptr;
        const size_t g2_sets = p.uni ? S : S_gen;              // (one pass: a record per set of that pass)
        p.g2_rec = w.take<u64>(g2 ? g2_sets * 65u : 0);
        g2_bytes = g2 ? g2_sets * 65u * sizeof(u64) : 0;
        if (!g2) p.g2_rec = nullptr;
        const size_t Tt = g2 ? 0 : T, Gt = g2 ? 0 : G;
        p.sum_tab = w.take<uint32_t>(Tt * 64); p.sum_valid = w.take<u64>(Tt); p.sum_le = w.take<int>(Tt);
        p.ent_tab = w.take<uint32_t>(Tt * 64); p.ent_valid = w.take<u64>(Tt); p.ent_le = w.take<int>(Tt);
        p.grp_tab = w.take<uint32_t>(Gt * 64); p.grp_valid = w.take<u64>(Gt); p.grp_le = w.take<int>(Gt);
        p.gent_tab = w.take<uint32_t>(Gt * 64); p.gent_le = w.take<int>(Gt);
        p.set_size = w.take<uint32_t>(lookback ? 0 : S); p.set_off = w.take<uint32_t>(lookback ? 0 : S);
        p.scratch = w.take<uint8_t>(lookback ? ((size_t)p.pool_slots + 1u) * p.set_stride : S * p.set_stride);
        uint8_t* const alt = w.take<uint8_t>(pingpong ? zero_bytes : 0);          // the second region (256-byte aligned like the first)
        if (!pass) { int rc = c->enc_ws.reserve(w.off + 256); if (rc) return rc; }
        else {
            uint8_t* mine = (uint8_t*)c->enc_ws.base;
            if (pingpong) {
                zero_other = alt; zero_len = zero_bytes;
                if (c->enc_parity) {                       // this call's turn on the second region: everything carved from the first moves over
                    const ptrdiff_t d = alt - mine;
                    auto over = [d](auto*& q) { q = reinterpret_cast<std::remove_reference_t<decltype(q)>>(reinterpret_cast<uint8_t*>(q) + d); };
                    over(p.status); over(p.ticket); over(p.err); over(p.need_generic); over(p.any_generic); over(p.status_gen); over(p.ticket_gen);
                    over(p.tree1); over(p.tree2); over(p.tree1_gen); over(p.tree2_gen); over(p.pool_map);
                    zero_other = mine; mine = alt;
                }
                c->enc_parity ^= 1;
                p.zero_next = reinterpret_cast<uint32_t*>(zero_other); p.zero_next_dwords = (uint32_t)(zero_bytes / 4u);
            }
            const bool zeroed = pingpong && zeroed_before.valid && zeroed_before.seq + 1 == ws_seq && zeroed_before.ptr == (void*)mine &&
                                zeroed_before.bytes == zero_bytes && zeroed_before.gen == c->enc_ws.gen;
            if (!zeroed) HIP_TRY(hipMemsetAsync(mine, 0, zero_bytes, st));   // look-back records, tickets, flags, pool map
            // the state look-back's granules are told apart by the call's number; zeroed only when they come to lie somewhere new
            // (another arena, another shape of call) or the number wraps
            c->enc_epoch = (c->enc_epoch + 1u) & 0x1FFFFFFFu;
            if (g2_bytes && (c->g2_zeroed_at != (void*)p.g2_rec || c->g2_zeroed_bytes != g2_bytes || c->g2_zeroed_gen != c->enc_ws.gen || c->enc_epoch == 0u)) {
                HIP_TRY(hipMemsetAsync(p.g2_rec, 0, g2_bytes, st));
                c->g2_zeroed_at = (void*)p.g2_rec; c->g2_zeroed_bytes = g2_bytes; c->g2_zeroed_gen = c->enc_ws.gen;
                if (c->enc_epoch == 0u) c->enc_epoch = 1u;
            }
            p.epoch = c->enc_epoch;
        }
    }
    // (a call that lays the workspace out WITHOUT state granules writes scratch, tables or summaries where an earlier call's granules lay:
    // the next call with granules must zero them again, whatever it finds at the same address)
    if (!g2_bytes) c->g2_zeroed_at = nullptr;
    p.out = (uint8_t*)d_streams; p.out_stride = stream_stride; p.out_len = d_stream_len;
    c->last_enc_err = p.err; c->last_enc_err2 = nullptr;
    if (c->timer.n > KernelTimer::kMax - 32) { HIP_TRY(hipStreamSynchronize(st)); timer_collect(c); }
    // (Running the placement passes of one sub-batch on a second stream beside the slab passes of the next was tried in round
    // 2: 4.996 vs 4.986 ms per 256 4K frames - the two kernels time-slice the CUs, nothing overlaps.)
    c->timer.mark(kT_begin, st);
    launch_encode(p, st, &c->timer);
    c->timer.mark(kT_enc_total, st);
    if (pingpong && zero_other) { c->prezero.ptr = zero_other; c->prezero.bytes = zero_len; c->prezero.gen = c->enc_ws.gen; c->prezero.seq = ws_seq; c->prezero.valid = true; }
    if (c->enc_adapt && place == 1) {                       // what this batch's streams look like, for the next call's set size (see above)
        if (hipMemcpyAsync(&c->host_word[12], d_stream_len, sizeof(uint32_t), hipMemcpyDeviceToHost, st) == hipSuccess) c->enc_hint_npx = p.npx;
        if (hipMemcpyAsync(&c->host_word[13], p.any_generic, sizeof(uint32_t), hipMemcpyDeviceToHost, st) == hipSuccess) c->enc_hint_images = (uint32_t)n_images;    // ... and how many flagged (flat) images it held: the grids of the next call's passes
    }
    c->last_enc.px = d_pixels; c->last_enc.ps = pixel_stride; c->last_enc.desc = *desc; c->last_enc.n = n_images;
    c->last_enc.out = d_streams; c->last_enc.os = stream_stride; c->last_enc.len = d_stream_len; c->last_enc.st = stream; c->last_enc.valid = true;
    if (const char* dump = c->enc_debug_dump.empty() ? nullptr : c->enc_debug_dump.c_str()) {          // diagnostics: the entry-state arrays of this call, raw
        (void)hipStreamSynchronize(st);
        if (FILE* fo = fopen(dump, "wb")) {
            auto put = [&](const void* d, size_t bytes) { std::vector<uint8_t> h(bytes); (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost); fwrite(h.data(), 1, bytes, fo); };
            const uint64_t hdr[4] = {T, G, (uint64_t)p.spi, (uint64_t)p.gpi};
            fwrite(hdr, 8, 4, fo);
            put(p.sum_tab, T * 256); put(p.sum_valid, T * 8); put(p.ent_tab, T * 256); put(p.ent_valid, T * 8);
            put(p.grp_tab, G * 256); put(p.grp_valid, G * 8); put(p.gent_tab, G * 256);
            fclose(fo);
        }
    }
    HIP_TRY(hipGetLastError());
    return QOIMI_OK;
}

// Differently shaped images in one call (qoibench.c:491-555 walks a directory): per-image descriptors and offsets.  The images are
// grouped by channel count (the kernels are compiled per count) and every group is placed order-free - each set parks its bytes in a
// scratch slot of its own, enc_offsets + enc_compact move them - so no set waits for another and any mix of sizes will do.
extern "C" int qoimi_encode_images(qoimi_ctx* c, const void* d_pixels, const size_t* pixel_offsets, const qoi_desc* descs, int n_images,
                                   void* d_streams, const size_t* stream_offsets, int* d_stream_len, void* stream) {
    if (!c || !d_pixels || !pixel_offsets || !descs || !d_streams || !stream_offsets || !d_stream_len || n_images <= 0) return fail(QOIMI_E_ARG, "NULL/empty argument");
    for (int i = 0; i < n_images; ++i) if (!desc_ok(&descs[i])) return fail(QOIMI_E_ARG, "descriptor rejected (qoi.h:364-372 rules)");
    DeviceGuard guard(c->device);
    ++c->enc_ws_seq; c->prezero.valid = false;               // (the encode workspace is laid out anew below: nothing a batch call zeroed ahead survives)
    c->g2_zeroed_at = nullptr;                               // ... nor the state granules of an earlier batch call
    hipStream_t st = (hipStream_t)stream;
    if (c->recheck_pending && hipStreamQuery(c->own_stream) == hipSuccess) {                  // (the repeat of the LDS-order self-test: see qoimi_encode_batch)
        c->recheck_pending = false;
        if (c->host_word[8] != 0u || c->test_force_recheck_fail) {
            c->xchg_ordered = false;
            c->enc_suspect_calls += c->enc_calls - c->enc_calls_last_passed;
            c->enc_calls_last_passed = c->enc_calls;
            c->recheck_failed_unreported = true;
        } else c->enc_calls_last_passed = c->enc_calls_at_check;
    }
    ++c->enc_calls;
    c->last_enc.valid = false;
    c->last_enc_err = nullptr; c->last_enc_err2 = nullptr;
    // the table of a channel group travels through pinned staging; both groups' tables and workspaces live side by side in the
    // arena (the second group's launches follow the first's on the stream and must not overwrite what those still read)
    size_t ws_off = 0;
    for (int pass = 0; pass < 2; ++pass) {               // pass 0 measures the arena (both groups), pass 1 carves and launches
        ws_off = 0;
        size_t pin_off = 0;
        for (uint32_t ch = 3; ch <= 4; ++ch) {
            std::vector<EncImage> tab;
            std::vector<int> who;
            uint64_t slabs = 0, groups = 0, units = 0;
            for (int i = 0; i < n_images; ++i) {
                if (descs[i].channels != ch) continue;
                EncImage e; memset(&e, 0, sizeof e);
                e.pixel_off = pixel_offsets[i]; e.out_off = stream_offsets[i];
                e.npx = descs[i].width * descs[i].height;
                e.spi = (e.npx + kEncSlabPx - 1u) / kEncSlabPx; e.gpi = (e.spi + 63u) / 64u;
                e.width = descs[i].width; e.height = descs[i].height; e.colorspace = descs[i].colorspace;
                e.slab_base = (uint32_t)slabs; e.grp_base = (uint32_t)groups;
                slabs += e.spi; groups += e.gpi;
                e.len_index = (uint32_t)i;
                tab.push_back(e); who.push_back(i);
            }
            if (tab.empty()) continue;
            if (slabs > 0xFFFFFFF0ull) return fail(QOIMI_E_ARG, "batch too large (slab index overflows 32 bits)");
            const uint32_t n = (uint32_t)tab.size();
            const uint32_t r = c->enc_set_slabs > 0 ? (uint32_t)(c->enc_set_slabs > (int)kEncMaxSetSlabs ? kEncMaxSetSlabs : c->enc_set_slabs)
                                                     : (slabs >= 3u * 65536u ? 3u : (slabs >= 16384u ? 2u : 1u));
            uint64_t sets = 0;
            for (EncImage& e : tab) { e.sets = (e.spi + r - 1u) / r; e.set_base = (uint32_t)sets; e.unit_base = (uint32_t)units; sets += e.sets; units += (e.sets + 3u) / 4u; }
            EncImage tail; memset(&tail, 0, sizeof tail);
            tail.set_base = (uint32_t)sets; tail.slab_base = (uint32_t)slabs; tail.grp_base = (uint32_t)groups; tail.unit_base = (uint32_t)units;
            tab.push_back(tail);
            EncParams p; memset(&p, 0, sizeof p);
            p.pixels = (const uint8_t*)d_pixels; p.out = (uint8_t*)d_streams; p.n_images = n; p.channels = (uint8_t)ch;
            p.set_slabs = r; p.set_px = r * kEncSlabPx; p.set_stride = r * kEncSlabWorst + 16u;
            p.probe_xchg = c->xchg_ordered ? 1 : 0; p.use_ticket = 0; p.warm = c->enc_warm ? 1 : 0; p.lookback = 0; p.pool = 0; p.spin_bound = 1u << 22; p.gen_slabs = kEncGenSetSlabs;
            const size_t T = (size_t)slabs, G = (size_t)groups, S = (size_t)sets;
            Carver w(pass ? (uint8_t*)c->enc_ws.base + ws_off : nullptr);
            if (!pass) w.base = nullptr;
            p.status = w.take<u64>(0); p.ticket = w.take<uint32_t>(n); p.err = w.take<uint32_t>(1);
            p.need_generic = w.take<uint32_t>(n); p.any_generic = w.take<uint32_t>(1);
            const size_t zero_bytes = w.off;
            EncImage* d_tab = w.take<EncImage>(tab.size());
            p.sum_tab = w.take<uint32_t>(T * 64); p.sum_valid = w.take<u64>(T); p.sum_le = w.take<int>(T);
            p.ent_tab = w.take<uint32_t>(T * 64); p.ent_valid = w.take<u64>(T); p.ent_le = w.take<int>(T);
            p.grp_tab = w.take<uint32_t>(G * 64); p.grp_valid = w.take<u64>(G); p.grp_le = w.take<int>(G);
            p.gent_tab = w.take<uint32_t>(G * 64); p.gent_le = w.take<int>(G);
            p.set_size = w.take<uint32_t>(S); p.set_off = w.take<uint32_t>(S);
            p.scratch = w.take<uint8_t>(S * p.set_stride);
            const size_t used = (w.off + 255u) & ~(size_t)255u;
            if (pass) {
                const size_t tbytes = tab.size() * sizeof(EncImage);
                if (hipMemsetAsync((uint8_t*)c->enc_ws.base + ws_off, 0, zero_bytes, st) != hipSuccess) return fail(QOIMI_E_INTERNAL, "hipMemsetAsync failed");
                memcpy((uint8_t*)c->enc_pin_buf + pin_off, tab.data(), tbytes);
                HIP_TRY(hipMemcpyAsync(d_tab, (uint8_t*)c->enc_pin_buf + pin_off, tbytes, hipMemcpyHostToDevice, st));
                HIP_TRY(hipEventRecord(c->enc_pin_ev, st));
                p.img_tab = d_tab; p.out_len = d_stream_len;           // (written at EncImage::len_index: the caller's image number)
                launch_encode_mixed(p, (uint32_t)units, (uint32_t)slabs, (uint32_t)groups, (uint32_t)sets, st, &c->timer);
                c->last_enc_err2 = c->last_enc_err; c->last_enc_err = p.err;       // (qoimi_encode_status looks at both channel groups)
            }
            ws_off += used;
            pin_off += (tab.size() * sizeof(EncImage) + 255u) & ~(size_t)255u;
        }
        if (!pass) {
            int rc = c->enc_ws.reserve(ws_off + 256); if (rc) return rc;
            // The staging of the previous call's tables may still be read by its copies - on whatever stream that call ran: wait for the
            // event recorded behind them (not for the stream: the call stays asynchronous) before the buffer is overwritten or freed.
            const size_t need = (size_t)(n_images + 2) * sizeof(EncImage) + 1024u;
            if (!c->enc_pin_ev) HIP_TRY(hipEventCreateWithFlags(&c->enc_pin_ev, hipEventDisableTiming));
            else HIP_TRY(hipEventSynchronize(c->enc_pin_ev));
            if (need > c->enc_pin_cap) {
                if (c->enc_pin_buf) (void)hipHostFree(c->enc_pin_buf);
                c->enc_pin_buf = nullptr; c->enc_pin_cap = 0;
                HIP_TRY(hipHostMalloc(&c->enc_pin_buf, need + 4096));
                c->enc_pin_cap = need + 4096;
            }
        }
    }
    HIP_TRY(hipGetLastError());
    return QOIMI_OK;
}

// Synchronise `stream` and report whether the last encode on this context tripped a
// device-side liveness bound (look-back spin limit).  Never expected; outputs of such a
// call must be discarded.
extern "C" int qoimi_encode_status(qoimi_ctx* c, void* stream) {
    if (!c) return fail(QOIMI_E_ARG, "ctx is NULL");
    DeviceGuard guard(c->device);
    HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
    if (c->last_enc.valid && c->last_enc.st != stream) HIP_TRY(hipStreamSynchronize((hipStream_t)c->last_enc.st));   // the stream the call was made on
    if (c->recheck_failed_unreported) {
        c->recheck_failed_unreported = false;
        return fail(QOIMI_E_INTERNAL, "the LDS exchange-order self-test failed on repetition: " + std::to_string(c->enc_suspect_calls) +
                    " earlier encode calls of this context are suspect (qoimi_encode_suspect_calls); the context now uses the order-free probe");
    }
    if (!c->last_enc_err) return QOIMI_OK;
    uint32_t err = 0, err2 = 0;
    HIP_TRY(hipMemcpy(&err, c->last_enc_err, sizeof err, hipMemcpyDeviceToHost));
    if (c->last_enc_err2) { HIP_TRY(hipMemcpy(&err2, c->last_enc_err2, sizeof err2, hipMemcpyDeviceToHost)); err |= err2; }
    if (err && c->last_enc.valid && c->enc_lookback != 0) {
        // A placement wait gave up (never observed: the sets a wait is for are resident or done unless another stream's launch holds
        // them out) or the scratch pool ran dry: the call is encoded again ORDER-FREE - no set waits for another, every set has a
        // scratch slot of its own - from the caller's buffers, which it has not read yet (it is asking for the status first), on the
        // stream the call was made on (the one this function waits for next, whatever `stream` is).
        const int forced = c->enc_lookback;
        c->enc_lookback = 0;
        c->last_enc.valid = false;
        const auto again = c->last_enc;
        const int rc = qoimi_encode_batch(c, again.px, again.ps, &again.desc, again.n, again.out, again.os, again.len, again.st);
        c->enc_lookback = forced;
        if (rc != QOIMI_OK) return rc;
        HIP_TRY(hipStreamSynchronize((hipStream_t)again.st));
        HIP_TRY(hipMemcpy(&err, c->last_enc_err, sizeof err, hipMemcpyDeviceToHost));
        c->enc_retries += 1;
    }
    if (err) return fail(QOIMI_E_INTERNAL, (err & 2u) ? "encode scratch pool exhausted" : "encode look-back exceeded its spin bound");
    return QOIMI_OK;
}

// ------------------------------------------------------------------------------------
// decode
// ------------------------------------------------------------------------------------
// segment size of a decode call (see the cost model below)
static uint32_t choose_seg_bytes(const qoimi_ctx* c, const int* sizes, const qoi_desc* descs, int n_images, bool honour_forced = true) {
    uint32_t B = honour_forced ? c->seg_bytes : 0u;         // (QOIMI_SEG_BYTES)
    if (B == 0 && c->dec_run_desc && c->dec_flat_seg) {
        // A call of FLAT images only (run descriptors): a lane's walk over its segment no longer writes the segment's pixels, it costs
        // its chunks alone - larger segments mean fewer entry states (780 bytes per segment whatever its size), fewer chances to miss
        // (a round per miss) and the same work.  The largest size that still gives 128 K lanes (1024 UI frames, 292 MB of streams: 512 /
        // 1024 / 2048 bytes = 10 / 5 / 3 rounds in 22.0 / 18.9 / 19.1 ms, profiles/r05_s6_dec_span.txt; 4096: P3 and P4 run short of lanes).
        bool all_flat = true;
        uint64_t bytes = 0;
        for (int i = 0; i < n_images && all_flat; ++i) {
            all_flat = sizes[i] > 22 && descs[i].width != 0 && dec_image_is_flat((uint32_t)sizes[i] - 8u, (uint32_t)((uint64_t)descs[i].width * descs[i].height));
            bytes += (uint64_t)(sizes[i] > 0 ? sizes[i] : 0);
        }
        // (round 6: not below 512 bytes - 128 UI frames, 36 MB of streams, took 256 and with it rounds that re-open nearly everything: the
        // stall rule sent them to the sequential pass, 147 ms where 512-byte segments take 5.2, profiles/r06_s9_uiflat_mid_batch.txt; a call
        // with streams for a quarter of those lanes still takes 512, smaller ones the general model)
        if (all_flat) {
            for (uint32_t cand = 4096u; cand >= 512u; cand >>= 1)
                if (bytes / cand >= 131072u) { B = cand; break; }
            if (B == 0 && bytes / 512u >= 32768u) B = 512u;
        }
    }
    if (B == 0) {
        // One lane decodes one segment.  Two costs pull in opposite directions (constants measured on MI355X):
        //   * a lane walks its segment serially, ~0.6 us per chunk-step over the four passes, and the two
        //     table-bound passes hold ~98 K lanes at a time: t_walk ~ B/1.2 * 0.6 us * ceil(lanes / 98304)
        //   * the per-image chains (S1/S2/S3 level 2) walk the image's 64-segment groups, ~0.5 us per group over
        //     the three chains, 16 wavefronts per image in two sweeps plus a 16-step hand-over:
        //     t_chain ~ (groups / 8 + 16) * 0.5 us with groups = largest stream / B / 64
        // Small batches therefore get small segments (more lanes), a single large image not too small ones.
        // Large batches end at 4 KiB: the 520-byte symbolic summary and the two 260-byte entry states per segment are then an
        // eighth of the stream (256 x 4K photographs: decode 10.3 ms at 2 KiB, 9.9 at 4 KiB - P3 -10 %, S3 halved; at 8 KiB the
        // transcoder's 64 lanes read 512 KiB apart and lose 15 %).
        uint64_t bytes = 0, largest = 0;
        for (int i = 0; i < n_images; ++i) {
            const uint64_t sz = (uint64_t)(sizes[i] > 0 ? sizes[i] : 0);
            bytes += sz; if (sz > largest) largest = sz;
        }
        double best = 1e30;
        for (uint32_t cand = 128; cand <= 4096u; cand <<= 1) {
            const double lanes = (double)bytes / cand;
            const double rounds = lanes <= 98304.0 ? 1.0 : lanes / 98304.0;
            double t = (cand / 1.2) * 0.6 * rounds + ((double)largest / cand / 64.0 / 8.0 + 16.0) * 0.5;
            if (n_images > 4) {
                // Batches (round 6, fitted to 8 .. 256 4K frames of photographs and sprites at every size, profiles/r06_s31_batch_by_seg.txt):
                // the passes run at the chip's throughput, ~5 us per MB of streams, plus the per-segment state - (1 + 140 / B) - and end
                // with the longest lane's walk, which grows with the segment: ~0.3 us per byte (photographs 0.1, sprites with long runs
                // 0.65); a call behind one that needed repair rounds counts 1.3 (a round's passes serve few segments: each is as long as
                // one walk).  sqrt(bytes): 512 bytes for 8 photographs, 1 KiB for 32, 2 KiB for 128 .. 256, 4 KiB from ~3.6 GB of streams.
                // (The model above it ties all sizes once the chip is full and took the largest: 32 sprite frames 5.5 ms at 4 KiB, 3.9 at 1 KiB.)
                const double kappa = c->dec_nonflat_repair ? 1.3 : 0.3;
                t = (double)bytes * 5e-6 * (1.0 + 140.0 / cand) + kappa * cand + ((double)largest / cand / 64.0 / 8.0 + 16.0) * 0.5;
            }
            if (t < best) { best = t; B = cand; }
        }
        // Calls of a few images whose streams are small: the chip is not full at 128 bytes (a 1080p photograph: 20 K segments, 320
        // wavefronts for 1024 SIMDs), a pass is as long as one lane's walk - shorter segments, two transcoder lanes each, as long as
        // the call stays below ~48 K segments (1280 x 720: 111 -> 100 us at 64 bytes, 1080p 120 -> 114, 1440p 132 -> 128 at 96; a 4K
        // photograph keeps 128: 168 us at 112, profiles/r06_s22_single_small_seg.txt).  No piece parse below 128 bytes: a call whose
        // transcoder cannot synchronise every segment takes the full five-phase parse.
        if (B == 128u && n_images <= 4 && c->dec_fused && c->dec_fine && c->dec_split && c->dec_small_seg && !c->dec_few_syncfail) {
            const uint64_t want = (bytes / 49152u + 15u) / 16u * 16u;
            B = want < 64u ? 64u : want < 128u ? (uint32_t)want : 128u;
        }
        // A call that MIXES flat images with others (a directory of screenshots and photographs, bench.py "mixed_directory"): the flat
        // ones' streams are a few hundred KB - a few dozen lanes at the 4 KiB the photographs' bytes ask for - and the symbolic pass walks
        // them several times (refinement passes): 4.8 of that leg's 9.3 ms.  Not above 1 KiB then (photographs lose a few per cent, 4 x
        // the lanes for the flat images' passes).
        if (B > 1024u && c->dec_run_desc)
            for (int i = 0; i < n_images; ++i)
                if (sizes[i] > 22 && descs[i].width != 0 && dec_image_is_flat((uint32_t)sizes[i] - 8u, (uint32_t)((uint64_t)descs[i].width * descs[i].height))) { B = 1024u; break; }
    }
    return B;
}

// one sub-batch: everything of qoimi_decode_batch for images whose record arena fits dec_rec_cap
static int decode_some(qoimi_ctx* c, const void* d_streams, size_t stream_stride,
                       const int* sizes, const qoi_desc* descs, int n_images, int channels,
                       void* d_pixels, size_t pixel_stride, void* stream, uint32_t B, bool lone_image, long long stats[4], const int* place = nullptr) {
    // place (a call decoded class by class): image i of this sub-call is the caller's image place[i] - its stream at place[i] * stream_stride,
    // its pixels at place[i] * pixel_stride; sizes / descs are the sub-call's own arrays
    int och = 0;
    std::vector<DecImage> imgs((size_t)n_images);
    uint64_t total = 0, total_g = 0, flat_total = 0;
    // Calls of a few images take the single-pass look-back kernel for pixel offsets and speculated slots (dec_scan_entry: one launch
    // where the three-level chains take ten); every image then begins on a multiple of kScanSegs segments.  Needs dec_transcode<0> (the
    // 128-byte piece parse's segment sizes) and falls back to the chains by itself where that pass cannot synchronise every segment.
    // (segments below 128 bytes, any multiple of 16 from 64 on: two transcoder lanes per segment; no piece parse for those - a call whose
    // transcoder cannot synchronise every segment takes the full five-phase parse)
    const bool small_seg = B >= 64u && B < 128u && B % 16u == 0u && c->dec_split;
    // (the context's previous call of a few images could not synchronise every segment - sprites with many alpha levels, noise - and paid for the
    // attempt: a wait, the parse, everything again through the chains.  The next such call takes the chains at once - and run descriptors for
    // long runs, a launch more on a path that no longer counts them; a call that synchronises everything switches back.  A lone 4K sprite
    // frame: 587 -> 404 us, profiles/r06_s34_single_kinds.txt; since the second sync run-up of dec_transcode<0> only streams built against the synchronisation get here)
    const bool skip_fused = n_images <= 4 && c->dec_few_syncfail && c->dec_fused_adapt;
    const bool fused_layout = c->dec_fused && !skip_fused && n_images <= 4 && c->dec_fine &&
                              (small_seg || (B % 128u == 0u && B / 128u >= 1u && B / 128u <= 64u && ((B / 128u) & (B / 128u - 1u)) == 0u));
    for (int i = 0; i < n_images; ++i) {
        if (sizes[i] < kHeaderBytes + kTrailerBytes) return fail(QOIMI_E_ARG, "stream shorter than 22 bytes (qoi.h:500)");
        if (!desc_ok(&descs[i])) return fail(QOIMI_E_ARG, "descriptor rejected (qoi.h:513-521 rules)");
        const int o = channels ? channels : descs[i].channels;
        if (och && o != och) return fail(QOIMI_E_ARG, "all images of a batch must share the output channel count");
        och = o;
        const size_t npx = (size_t)descs[i].width * descs[i].height;
        if (npx * (size_t)o > pixel_stride) return fail(QOIMI_E_ARG, "pixel_stride smaller than a decoded image");
        if ((size_t)sizes[i] > stream_stride && !lone_image) return fail(QOIMI_E_ARG, "stream longer than stream_stride");
        DecImage& im = imgs[(size_t)i];
        memset(&im, 0, sizeof im);
        im.stream_off = (size_t)(place ? place[i] : i) * stream_stride;
        im.out_index = (uint32_t)(place ? place[i] : i);
        im.chunks_end = (uint32_t)(sizes[i] - kTrailerBytes);
        im.npx = (uint32_t)npx;
        if (fused_layout) { total = (total + kScanSegs - 1u) / kScanSegs * kScanSegs; total_g = total / 64u; }
        im.seg_base = (uint32_t)total;
        im.nseg = (im.chunks_end - kHeaderBytes + B - 1u) / B;
        im.grp_base = (uint32_t)total_g;
        im.ngrp = (im.nseg + 63u) / 64u;
        im.desc_base = kNoRunDesc;
        if (c->dec_run_desc && im.nseg != 0u && dec_image_is_flat(im.chunks_end, im.npx)) { im.desc_base = (uint32_t)flat_total; flat_total += im.nseg; }
        total += im.nseg;
        total_g += im.ngrp;
    }
    if (fused_layout) { total = (total + kScanSegs - 1u) / kScanSegs * kScanSegs; total_g = total / 64u; }
    if (total > 0xFFFFFFF0ull) return fail(QOIMI_E_ARG, "batch too large (segment index overflows 32 bits)");
    DeviceGuard guard(c->device);
    hipStream_t st = (hipStream_t)stream;
    // The previous call of this context may have returned on its pinned result words while its dec_fill was still retiring (it zeroes the
    // counter header last).  On the same stream this call's work is ordered behind it; a caller that changes streams gets the wait here.
    if (c->dec_tail_open && c->dec_tail_stream != stream) HIP_TRY(hipStreamSynchronize((hipStream_t)c->dec_tail_stream));
    c->dec_tail_open = false;

    DecParams p;
    memset(&p, 0, sizeof p);
    bool fused = fused_layout && total != 0;
    p.streams = (const uint8_t*)d_streams; p.n_images = (uint32_t)n_images;
    p.total_segs = (uint32_t)total; p.total_grps = (uint32_t)total_g; p.seg_bytes = B;
    p.rec_rows = rec_rows_of(B);
    if (fused && B <= (uint32_t)c->dec_split_max && c->dec_split) {          // two transcoder lanes per segment (dec_transcode<0, .., SPLIT>): rows for two halves
        p.tr_split = 1u; p.tr_rows_half = rec_rows_of(B / 2u); p.rec_rows = 2u * p.tr_rows_half;
        p.tr_scan = c->dec_tr_scan ? 1u : 0u;
    }
    p.flat_segs = (uint32_t)flat_total;
    p.desc_cap = (p.tr_split ? 2u * rec_max_records(B / 2u) : rec_max_records(B)) / 2u + 2u;              // a run ends with the record behind it: every second record at most
    // descriptors for the long runs of the other images as well - not for calls of a few images without a flat one (one more launch
    // on a path that counts them)
    // (... nor for a call of a few images unless the context's previous one met long runs by the thousand - a sprite's transparent bands: its P4 is
    // then as long as the lane that writes a band 16 bytes at a time, 257 us for a 4K frame against 112 with descriptors)
    p.desc_all = (c->dec_run_desc >= 2 && (n_images > 4 || flat_total != 0 || skip_fused || c->dec_few_longruns)) ? 1u : 0u;
    p.sync_all = 0;
    p.p3_plain = (uint32_t)c->dec_p3_plain;
    p.refine_inner = (uint32_t)c->dec_inner;
    {   // extra first-round passes only if the call holds a flat image at all
        bool any_flat = false;
        for (int i = 0; i < n_images && !any_flat; ++i) any_flat = sizes[i] > 22 && dec_image_is_flat((uint32_t)sizes[i] - 8u, descs[i].width * descs[i].height);
        p.first_inner = any_flat ? (uint32_t)c->dec_inner1 : 0u;
    }
    p.pixels = (uint8_t*)d_pixels; p.pixel_stride = pixel_stride;
    const size_t Q = total + 1;   // +1: check of segment q reads entry[q+1]
    {   // P1/P2 on 128-byte pieces when a segment is 1, 2, 4 ... 64 of them
        const uint32_t g = B / 128u;
        const bool ok = B % 128u == 0u && g >= 1u && g <= 64u && (g & (g - 1u)) == 0u && c->dec_fine;
        p.fine_per_seg = ok ? g : 0u;
        p.fine_shift = 0;
        while (ok && (1u << p.fine_shift) < g) ++p.fine_shift;
        if ((uint64_t)total * (p.fine_per_seg ? p.fine_per_seg : 1u) > 0xFFFFFF00ull) return fail(QOIMI_E_ARG, "batch too large (piece index overflows 32 bits)");
        p.sync_all = p.fine_per_seg ? 0u : 1u;     // no piece parse for this segment size: full parse, then transcode from S1's phases
    }
    {   // a few large images: the per-image level of the state chain as several workgroups per image (dec_chain_state_l2m)
        uint64_t most = 0;
        for (const DecImage& im : imgs) most = im.ngrp > most ? im.ngrp : most;
        p.l2_wgs = (c->dec_l2_wgs && n_images <= 4 && (most >= 128u || c->dec_l2_wgs == 2)) ? 8u : 1u;                    // (4 images x 8 flags fit the counter header)
    }
    for (int pass = 0; pass < 2; ++pass) {
        Carver w(pass ? c->dec_ws.base : nullptr);
        p.pending = w.take<uint32_t>(4); p.redo_segs = p.pending ? p.pending + 1 : nullptr; p.sync_fails = p.pending ? p.pending + 2 : nullptr;
        p.run_queue_n = p.pending ? p.pending + 3 : nullptr;
        p.l2_ticket = p.pending ? p.pending + 8 : nullptr; p.l2_flag = p.pending ? p.pending + 16 : nullptr;       // words 8..11 and 16..47 of the zeroed 256-byte header
        p.conv = (p.pending && c->dec_conv) ? p.pending + 48 : nullptr;                                      // words 48..63: refinement passes that changed something (DecParams::conv)
        p.images = w.take<DecImage>((size_t)n_images);
        p.first_bad = w.take<uint32_t>((size_t)n_images);
        p.parse = w.take<ParseRec>(Q); p.entry_phase = w.take<uint8_t>(Q); p.px_off = w.take<uint32_t>(Q);
        p.slot_rec = w.take<SlotRec>(Q); p.slot_in = w.take<uint8_t>(Q); p.alpha_in = w.take<uint8_t>(Q);
        p.summary = w.take<u64>(Q * 65); p.entry = w.take<uint32_t>(Q * 65); p.fix = w.take<uint32_t>(Q * 65);
        const size_t NG = total_g + 1;
        p.grp_parse = w.take<ParseRec>(NG); p.grp_phase = w.take<uint8_t>(NG); p.grp_off = w.take<uint32_t>(NG);
        p.grp_slot = w.take<SlotRec>(NG); p.grp_slot_in = w.take<uint8_t>(NG); p.grp_alpha_in = w.take<uint8_t>(NG);
        p.grp_summary = w.take<u64>(NG * 65); p.grp_entry = w.take<uint32_t>(NG * 65);
        p.l2_sum = w.take<u64>((size_t)n_images * p.l2_wgs * 65);
        // calls of a few images: four wavefronts per group in the state chain (quarter summaries), prefixes instead of a chain of workgroups
        // at the per-image level
        const bool few = n_images <= 4 && c->dec_fused != 0;
        p.qtr_summary = w.take<u64>(few ? NG * 4u * 65u : 0);
        p.grp_prefix = w.take<u64>(few ? NG * 65u : 0); p.share_prefix = w.take<u64>(few ? (size_t)n_images * 8u * 16u * 65u : 0);
        if (few && p.l2_wgs < 8u) p.l2_sum = w.take<u64>((size_t)n_images * 8u * 65u);                  // (l2_sum above was sized for l2_wgs workgroups)
        p.s3_ctr = w.take<uint32_t>(few ? (size_t)n_images * 8u * 17u : 0); p.share_sum = w.take<u64>(few ? (size_t)n_images * 8u * 16u * 65u : 0);
        if (!few || !c->dec_s3_ride) { p.s3_ctr = nullptr; }
        if (!few) { p.qtr_summary = nullptr; p.grp_prefix = nullptr; p.share_prefix = nullptr; p.share_sum = nullptr; }
        p.rec_gran = w.take<uint32_t>(Q);
        p.run_cnt = w.take<uint32_t>((flat_total || p.desc_all) ? Q : 0);
        p.run_queue = w.take<uint32_t>((flat_total || p.desc_all) ? Q : 0);
        p.run_desc = w.take<uint4>((size_t)flat_total * p.desc_cap);
        p.sync_fail = w.take<uint8_t>(Q);
        p.recs = w.take<uint32_t>(((Q + 63u) / 64u) * p.rec_rows * 256u);
        if (!pass) { int rc = c->dec_ws.reserve(w.off + 256); if (rc) return rc; }
    }
    if (fused) {
        const size_t words = (size_t)(total / (kScanSegs / 2u)) + 64u;      // (a word per 128 segments where the scan rides on the two-lane transcoder)
        const unsigned gen = c->dec_scan.gen;
        if (c->dec_scan.reserve(words * sizeof(u64)) != QOIMI_OK) { fused = false; p.tr_split = 0u; p.tr_scan = 0u; }          // (no memory for a few KB: the chains will do)
        else {
            c->dec_epoch = (c->dec_epoch + 1u) & 0xFFFFu;
            if (gen != c->dec_scan.gen || c->dec_epoch == 0u) {                           // a new arena, or the tag wraps: no word may carry a tag from before
                HIP_TRY(hipMemsetAsync(c->dec_scan.base, 0, c->dec_scan.cap, st));
                if (c->dec_epoch == 0u) c->dec_epoch = 1u;
            }
            p.fused = 1u; p.epoch = c->dec_epoch; p.scan_status = (u64*)c->dec_scan.base; p.scan_ticket = p.pending + 5;
            p.host_result = &c->host_word[20];
        }
    }
    // Calls of a few images whose predecessor on this context left the counter header zeroed (its dec_fill, see there): the table rides in
    // dec_transcode<0>'s kernel arguments - no copy at all in front of the first kernel.
    const bool hdr_clean = c->dec_hdr_zero.valid && c->dec_hdr_zero.at == (void*)p.pending && c->dec_hdr_zero.gen == c->dec_ws.gen;
    c->dec_hdr_zero.valid = false;
    if (fused && hdr_clean && n_images <= 4) {
        p.tab_in_args = 1u;
        for (int i = 0; i < n_images; ++i) p.tab4[i] = imgs[(size_t)i];
    } else
    {   // image table through pinned staging: no synchronisation (every decode call ends with one, so the staging buffer is free
        // again when the next call fills it).  The four counter words in front of it (pending, redo_segs, sync_fails: the
        // arena's first 256 bytes, the table follows them) travel zeroed in the same copy: no memset launches in round one.
        static_assert(sizeof(DecImage) % 8 == 0, "image table entries keep their alignment behind the counter words");
        const size_t bytes = 256u + imgs.size() * sizeof(DecImage);
        if ((uint8_t*)p.images != (uint8_t*)p.pending + 256u) return fail(QOIMI_E_INTERNAL, "decode workspace layout changed");
        if (bytes > c->pin_cap) {
            if (c->pin_buf) (void)hipHostFree(c->pin_buf);
            c->pin_buf = nullptr; c->pin_cap = 0;
            HIP_TRY(hipHostMalloc(&c->pin_buf, bytes + 4096));
            c->pin_cap = bytes + 4096;
        }
        memset(c->pin_buf, 0, 256);
        memcpy((uint8_t*)c->pin_buf + 256, imgs.data(), bytes - 256u);
        HIP_TRY(hipMemcpyAsync(p.pending, c->pin_buf, bytes, hipMemcpyHostToDevice, st));
    }

    if (fused) launch_decode_fused_front(p, st, &c->timer);
    else launch_decode_parse(p, st, &c->timer);
    long long rounds = 0, stats_seq = 0;
    // A round that re-opens nearly as many segments as the one before it is not getting anywhere (a stream built against the
    // speculation: one verified segment per image and round): two such rounds in a row and the rest goes to the sequential
    // pass at once instead of after dec_max_rounds relaunches over everything (redo_segs accumulates over the rounds).
    uint32_t redo_cum = 0, open_prev = 0xFFFFFFFFu; int stalled = 0;
    for (;;) {
        if (rounds > 0) HIP_TRY(hipMemsetAsync(p.pending, 0, sizeof(uint32_t), st));
        if (rounds > 0 && p.conv) HIP_TRY(hipMemsetAsync(p.conv, 0, 16 * sizeof(uint32_t), st));
        p.l2_tag_base = (uint32_t)rounds * 65536u + 1u;            // (a round launches S3 1 + first_inner / refine_inner times: far fewer than 65536)
        // the first round of a call of a few images: dec_fill leaves the round's counters in pinned host words (no copy back)
        p.tail_fused = (p.fused && rounds == 0) ? 1u : 0u;
        launch_decode_round(p, och, rounds > 0 && c->dec_refine, st, &c->timer);
        ++rounds;
        // pixels the chunks never reach (cheap; redone if the round has to be repeated) - before the read-back,
        // so that the one synchronisation per round also ends the call
        launch_decode_fill(p, och, st, &c->timer);
        c->timer.mark(kT_dec_total, st);
        if (!p.total_segs) { HIP_TRY(hipStreamSynchronize(st)); break; }
        if (p.tail_fused) {
            // dec_fill's first wavefront writes the round's counters into pinned words when everything in front of it - every pixel of the
            // call: dec_segments_rec has ended - is done.  Where no image needs filling (word 23) the call may return on seeing them: the
            // rest of that launch writes nothing.  A few microseconds earlier than the stream's completion signal; after 2 ms of looking (or
            // with per-kernel timing on) the stream is waited for as ever.
            volatile uint32_t* const hw = c->host_word;
            bool seen = false;
            if (!c->timer.on) {
                for (uint32_t spin = 0; spin < 400000u; ++spin) {
                    if (hw[24] == p.epoch) { seen = true; break; }
                    __builtin_ia32_pause();
                }
            }
            if (!seen || hw[23] != 0u) HIP_TRY(hipStreamSynchronize(st));
            else { c->dec_tail_open = true; c->dec_tail_stream = stream; }
            c->host_word[0] = c->host_word[20]; c->host_word[1] = c->host_word[21]; c->host_word[2] = c->host_word[22];
            c->dec_few_longruns = c->host_word[25] >= 1024u;
            // (that dec_fill left the header zeroed; good for the next call if nothing else of this call touches it: no further round)
            c->dec_hdr_zero.at = (void*)p.pending; c->dec_hdr_zero.gen = c->dec_ws.gen; c->dec_hdr_zero.valid = c->host_word[0] == 0u && c->host_word[2] == 0u;
        } else {
            HIP_TRY(hipMemcpyAsync(c->host_word, p.pending, 3 * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(hipStreamSynchronize(st));
        }
        timer_collect(c);
        if (p.fused && c->host_word[2] != 0u) {
            // dec_transcode<0> could not synchronise every segment (runs of equally long multi-byte chunks: noise): dec_scan_entry and
            // everything behind it returned at once.  The five-phase parse, the three-level chains and the round again, on the records
            // that stand (the flagged segments are transcoded by dec_transcode<1>).
            p.fused = 0u; fused = false; p.tr_scan = 0u;         // (tr_scan off: the kernels of the chains must not return on sync_fails)
            rounds = 0;
            if (p.fine_per_seg) launch_decode_parse_rest(p, st, &c->timer);
            else launch_decode_parse(p, st, &c->timer);           // (segment sizes without the piece parse: every segment again; sync_fails stands - the call's statistics)
            if (p.conv) HIP_TRY(hipMemsetAsync(p.conv, 0, 16 * sizeof(uint32_t), st));
            continue;
        }
        p.fused = 0u; p.tr_scan = 0u;                       // (rounds after a failed check are the three-level ones, from the image's first bad segment)
        if (c->host_word[0] == 0) break;
        {
            const uint32_t open_now = c->host_word[1] - redo_cum;
            redo_cum = c->host_word[1];
            // (a round that closes less than a 64th of what was open; round 5 asked for a 16th and sent UI frames at small segments - slow
            // but steady, a few per cent per round - to the sequential pass: 30 x the time of the rounds they still needed)
            stalled = (rounds >= 4 && (uint64_t)open_now * 64u > (uint64_t)open_prev * 63u) ? stalled + 1 : 0;
            open_prev = open_now;
        }
        if (rounds >= c->dec_max_rounds || stalled >= 2) {
            // bounded: whatever is still open is finished by the linear sequential pass (see dec_sequential)
            launch_decode_sequential(p, och, st, &c->timer);
            launch_decode_fill(p, och, st, &c->timer);
            c->timer.mark(kT_dec_total, st);
            HIP_TRY(hipStreamSynchronize(st));
            stats_seq = (long long)c->host_word[0];
            break;
        }
    }
    HIP_TRY(hipGetLastError());
    timer_collect(c);
    if (const char* dump = c->dec_debug_dump.empty() ? nullptr : c->dec_debug_dump.c_str()) {                // diagnostics: per-segment arrays of this call, raw
        (void)hipStreamSynchronize(st);
        if (FILE* fo = fopen(dump, "wb")) {
            auto put = [&](const void* d, size_t bytes) { std::vector<uint8_t> h(bytes); (void)hipMemcpy(h.data(), d, bytes, hipMemcpyDeviceToHost); fwrite(h.data(), 1, bytes, fo); };
            const uint64_t hdr[4] = {total, (uint64_t)p.tr_split, (uint64_t)p.rec_rows, (uint64_t)B};
            fwrite(hdr, 8, 4, fo);
            put(p.rec_gran, total * 4); put(p.parse, total * sizeof(ParseRec)); put(p.px_off, total * 4); put(p.sync_fail, total);
            fclose(fo);
        }
    }
    stats[0] = rounds;
    stats[1] = p.total_segs ? c->host_word[1] : 0;
    stats[2] = (long long)total;
    stats[3] = p.total_segs ? c->host_word[2] : 0;
    c->dec_seq_images += stats_seq;
    if (n_images <= 4 && p.total_segs) c->dec_few_syncfail = c->host_word[2] != 0u;
    return QOIMI_OK;
}

extern "C" int qoimi_decode_batch(qoimi_ctx* c, const void* d_streams, size_t stream_stride,
                                  const int* sizes, const qoi_desc* descs, int n_images, int channels,
                                  void* d_pixels, size_t pixel_stride, void* stream) {
    if (!c || !d_streams || !sizes || !descs || !d_pixels || n_images <= 0) return fail(QOIMI_E_ARG, "NULL/empty argument");
    if (channels != 0 && channels != 3 && channels != 4) return fail(QOIMI_E_ARG, "channels must be 0, 3 or 4 (qoi.h:499)");
    for (int i = 1; i < n_images && channels == 0; ++i)
        if (descs[i].channels != descs[0].channels) return fail(QOIMI_E_ARG, "all images of a batch must share the output channel count");
    // The chunk records take four bytes per stream byte (worst case) while a call is in flight.  Calls whose streams would
    // need more than dec_rec_cap are decoded as consecutive sub-batches of whole images through the same workspace.
    const uint64_t cap_stream = (uint64_t)(c->dec_rec_cap / 4u) - (uint64_t)(c->dec_rec_cap / 4u) / 64u;
    long long acc[4] = {0, 0, 0, 0};
    auto sub_batches = [&](const int* sz_v, const qoi_desc* ds_v, const int* place, int n_all, uint32_t B) -> int {
        for (int first = 0; first < n_all;) {
            uint64_t bytes = 0;
            int n = 0;
            while (first + n < n_all) {
                const uint64_t sz = (uint64_t)(sz_v[first + n] > 0 ? sz_v[first + n] : 0) + B;
                if (n > 0 && bytes + sz > cap_stream) break;
                bytes += sz; ++n;
            }
            long long st3[4] = {0, 0, 0, 0};
            int rc;
            if (place) rc = decode_some(c, d_streams, stream_stride, sz_v + first, ds_v + first, n, channels, d_pixels, pixel_stride, stream, B, false, st3, place + first);
            else rc = decode_some(c, (const uint8_t*)d_streams + (size_t)first * stream_stride, stream_stride, sz_v + first, ds_v + first, n, channels,
                                  (uint8_t*)d_pixels + (size_t)first * pixel_stride, pixel_stride, stream, B, n_images == 1, st3);
            if (rc != QOIMI_OK) return rc;
            acc[0] = st3[0] > acc[0] ? st3[0] : acc[0]; acc[1] += st3[1]; acc[2] += st3[2]; acc[3] += st3[3];
            first += n;
        }
        return QOIMI_OK;
    };
    // A call that MIXES flat images (UI frames, constant frames: streams of a few hundred KB) with others - a directory of screenshots and
    // photographs - is decoded CLASS BY CLASS: the flat images' passes (a few refinement passes in front of their P4, the P4 that leaves run
    // descriptors) are as long as one lane's walk over one segment, and the segment size the other images' bytes ask for made each of them
    // ~270 us for a few hundred lanes (3 of the mixed directory's 5.4 ms, profiles/r06_s28_mixed_timeline.txt).  Each class takes the
    // segment size of its own bytes; an image's place in the caller's buffers travels in the table (DecImage::out_index).
    int n_flat = 0;
    if (n_images > 4)
        for (int i = 0; i < n_images; ++i)
            n_flat += (sizes[i] > 22 && descs[i].width != 0 && dec_image_is_flat((uint32_t)sizes[i] - 8u, (uint32_t)((uint64_t)descs[i].width * descs[i].height))) ? 1 : 0;
    if (n_flat != 0 && n_flat != n_images && c->dec_run_desc && c->dec_class_split) {
        for (int cls = 0; cls < 2; ++cls) {
            std::vector<int> place, sz_v; std::vector<qoi_desc> ds_v;
            for (int i = 0; i < n_images; ++i) {
                const bool flat = sizes[i] > 22 && descs[i].width != 0 && dec_image_is_flat((uint32_t)sizes[i] - 8u, (uint32_t)((uint64_t)descs[i].width * descs[i].height));
                if ((flat ? 1 : 0) == cls) { place.push_back(i); sz_v.push_back(sizes[i]); ds_v.push_back(descs[i]); }
            }
            const uint32_t B = choose_seg_bytes(c, sz_v.data(), ds_v.data(), (int)place.size(), cls == 0);      // (QOIMI_SEG_BYTES: the other images' size; the flat class keeps its rule)
            const long long before = acc[0];
            acc[0] = 0;
            const int rc = sub_batches(sz_v.data(), ds_v.data(), place.data(), (int)place.size(), B);
            if (rc != QOIMI_OK) return rc;
            if (cls == 0 && place.size() > 4u) c->dec_nonflat_repair = acc[0] > 1;
            acc[0] = acc[0] > before ? acc[0] : before;
        }
    } else {
        const uint32_t B = choose_seg_bytes(c, sizes, descs, n_images);
        const int rc = sub_batches(sizes, descs, nullptr, n_images, B);
        if (rc != QOIMI_OK) return rc;
        if (n_images > 4 && n_flat == 0) c->dec_nonflat_repair = acc[0] > 1;
    }
    c->dec_stats[0] = acc[0]; c->dec_stats[1] = acc[1]; c->dec_stats[2] = acc[2]; c->dec_stats[3] = acc[3];
    return QOIMI_OK;
}

// ------------------------------------------------------------------------------------
// synthetic frames
// ------------------------------------------------------------------------------------
extern "C" int qoimi_synth_frames(qoimi_ctx* c, int kind, unsigned seed, unsigned first_frame,
                                  int n_frames, unsigned width, unsigned height,
                                  void* d_pixels, size_t pixel_stride, void* stream) {
    if (!c || !d_pixels || n_frames <= 0 || kind < 0 || kind > 5 || width == 0 || height == 0)
        return fail(QOIMI_E_ARG, "bad argument");
    const size_t npx = (size_t)width * height;
    if (npx >= kPixelCap || pixel_stride < npx * 4 || n_frames > 65535) return fail(QOIMI_E_ARG, "bad frame geometry");
    DeviceGuard guard(c->device);
    SynthParams p;
    p.pixels = (uint8_t*)d_pixels; p.pixel_stride = pixel_stride; p.npx = (uint32_t)npx; p.width = width;
    p.n_frames = (uint32_t)n_frames; p.first_frame = first_frame; p.seed = seed; p.kind = kind;
    launch_synth(p, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return QOIMI_OK;
}

extern "C" int qoimi_hash_streams(qoimi_ctx* c, const void* d_streams, size_t stream_stride, const int* d_stream_len, int n_streams,
                                  unsigned long long* d_hash, void* stream) {
    if (!c || !d_streams || !d_stream_len || !d_hash || n_streams <= 0) return fail(QOIMI_E_ARG, "NULL/empty argument");
    DeviceGuard guard(c->device);
    launch_hash_streams((const uint8_t*)d_streams, stream_stride, d_stream_len, (uint32_t)n_streams, (u64*)d_hash, (hipStream_t)stream);
    HIP_TRY(hipGetLastError());
    return QOIMI_OK;
}

// ------------------------------------------------------------------------------------
// Part 1 — drop-in entry points on host pointers
// ------------------------------------------------------------------------------------
// One context per CALLING THREAD (qoi.h:339,357-362,489-495: the reference keeps no state between calls and is callable
// from any number of threads at once).  Round 1 serialised every call on one global context behind a mutex; now a thread's
// calls run on its own context - own workspace, own non-blocking stream - so concurrent callers overlap their copies and
// kernels on the GPU.  A thread's context goes away with the thread.
// A fresh malloc of tens of megabytes is untouched address space: the copy back from the device would take a page fault every
// 4 KiB (2 ms of a 2.6 ms qoi_decode of a 4K frame, bench.py "dropin_host_pointers").  Ask for huge pages and have the range
// populated in one call instead; where the kernel knows neither, nothing is lost.
static void prefault_pages(void* p, size_t n) {
    if (n < ((size_t)1 << 20)) return;
    const uintptr_t a = ((uintptr_t)p + 4095u) & ~(uintptr_t)4095u, e = ((uintptr_t)p + n) & ~(uintptr_t)4095u;
    if (e <= a) return;
#ifdef MADV_HUGEPAGE
    (void)madvise((void*)a, e - a, MADV_HUGEPAGE);
#endif
#ifndef MADV_POPULATE_WRITE
#define MADV_POPULATE_WRITE 23
#endif
    (void)madvise((void*)a, e - a, MADV_POPULATE_WRITE);
}

// The helpers that populate qoi_decode's result pages while the stream goes in and the kernels run: TWO parked threads per
// calling thread, started at its first large decode and woken per call (round 2 created and joined two std::threads in every
// call).  Per 4K frame: no populate 2.6 ms, one thread 1.95, two 1.61, three 2.4, four 2.5 - they contend for the address-space lock.
class Prefaulter {
    static constexpr int kThreads = 2;
    std::thread th[kThreads];
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    void* ptr[kThreads] = {nullptr, nullptr};
    size_t len[kThreads] = {0, 0};
    unsigned long long ticket[kThreads] = {0, 0}, done[kThreads] = {0, 0};
    bool started = false, quit = false, no_helpers = false;
    void loop(int i) {
        unsigned long long seen = 0;
        for (;;) {
            void* p; size_t n;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return quit || ticket[i] != seen; });
                if (quit) return;
                seen = ticket[i]; p = ptr[i]; n = len[i];
            }
            prefault_pages(p, n);
            { std::lock_guard<std::mutex> lk(mu); done[i] = seen; }
            cv_done.notify_all();
        }
    }
public:
    // populate [p, p + n) in the background; wait() returns when it is done
    void start(void* p, size_t n) {
        if (!started) {
            if (no_helpers) { prefault_pages(p, n); return; }
            int made = 0;
            try { for (; made < kThreads; ++made) th[made] = std::thread(&Prefaulter::loop, this, made); started = true; }
            catch (...) {
                // no threads to be had: the ones that did start are told to quit and joined, the state stays "no helpers" for good
                // (a second attempt would assign to a joinable std::thread), and this call populates here
                { std::lock_guard<std::mutex> lk(mu); quit = true; }
                cv_work.notify_all();
                for (int i = 0; i < made; ++i) if (th[i].joinable()) th[i].join();
                no_helpers = true;
                prefault_pages(p, n);
                return;
            }
        }
        const size_t part = ((n / kThreads) + 4095u) & ~(size_t)4095u;
        std::lock_guard<std::mutex> lk(mu);
        for (int i = 0; i < kThreads; ++i) {
            const size_t lo = (size_t)i * part;
            ptr[i] = (uint8_t*)p + (lo < n ? lo : n);
            len[i] = lo >= n ? 0 : ((i == kThreads - 1 || lo + part > n) ? n - lo : part);
            ++ticket[i];
        }
        cv_work.notify_all();
    }
    void wait() {
        if (!started) return;
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { for (int i = 0; i < kThreads; ++i) if (done[i] != ticket[i]) return false; return true; });
    }
    ~Prefaulter() {
        if (!started) return;
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv_work.notify_all();
        for (auto& t : th) if (t.joinable()) t.join();
    }
};

struct ThreadCtx {
    qoimi_ctx* c = nullptr;
    bool tried = false;
    Prefaulter pf;
    ~ThreadCtx() { if (c) qoimi_ctx_destroy(c); }
};
static thread_local ThreadCtx t_ctx;
static std::mutex g_mutex;                 // guards the one-time warning only

static qoimi_ctx* thread_ctx() {
    if (!t_ctx.c && !t_ctx.tried) {
        t_ctx.tried = true;
        int dev = 0;
        if (const char* e = getenv("QOIMI_DEVICE")) dev = atoi(e);
        if (qoimi_ctx_create(dev, &t_ctx.c) != QOIMI_OK) {
            std::lock_guard<std::mutex> lock(g_mutex);
            fprintf(stderr, "qoi_mi355x: no usable MI355X (%s); there is no CPU fallback\n", t_error.c_str());
            t_ctx.c = nullptr;
        } else t_ctx.c->dropin = true;
    }
    return t_ctx.c;
}

extern "C" void* qoi_encode(const void* data, const qoi_desc* desc, int* out_len) {
    if (!data || !out_len || !desc_ok(desc)) return NULL;                 // qoi.h:364-372
    qoimi_ctx* c = thread_ctx();
    if (!c) return NULL;
    DeviceGuard guard(c->device);
    hipStream_t st = c->own_stream;
    const size_t npx = (size_t)desc->width * desc->height;
    const size_t in_bytes = npx * desc->channels;
    const size_t bound = qoimi_encode_bound(desc);                        // qoi.h:374-376
    if (c->io_a.reserve(in_bytes + 16) || c->io_b.reserve(bound + 16) || c->io_c.reserve(256)) return NULL;
    void* result = NULL;
    // The result is the reference's allocation (qoi.h:374-379: the worst case, w * h * (channels + 1) + 22 bytes) - a caller written
    // against the reference may count on that capacity.  Only the pages the stream will touch are populated (by the thread's parked
    // helpers, while the pixels go in and the kernels run); the rest of the allocation stays untouched address space.  What it costs:
    // a 4K frame's worst case is 39.6 MiB, above the 32 MiB ceiling of glibc's dynamic mmap threshold, so every call maps fresh
    // zero-filled pages and the caller's free() unmaps them (qoibench's encode-free loop, qoibench.c:446-449: 1.7 ms per call
    // against 0.84 with the tight buffer, profiles/r06_s1_dropin_worst_case.txt).  QOIMI_ENCODE_TIGHT_BUFFER=1 (opt-in, read when the
    // thread's context is created) sizes the buffer by this thread's previous stream instead (+ 1/8; a third of the bound at first; an
    // exact buffer in the rare case the stream turns out longer): it comes back from the allocator's heap with its pages in place.
    const bool worst_case = !c->tight_buffer;
    const size_t guess = worst_case ? bound : (c->last_drop_len ? c->last_drop_len + c->last_drop_len / 8u : bound / 3u);
    size_t ahead = guess < bound ? guess : bound;
    if (ahead < (size_t)kHeaderBytes + kTrailerBytes) ahead = (size_t)kHeaderBytes + kTrailerBytes;
    uint8_t* bytes = (uint8_t*)malloc(ahead);
    if (!bytes) return NULL;
    // The result's pages are populated by the thread's parked helpers WHILE the pixels go in and the kernels run (round 2
    // populated after the kernels: 0.35 ms of a 1.3 ms call).
    const size_t expect = worst_case ? (c->last_drop_len ? c->last_drop_len + c->last_drop_len / 8u : bound / 3u) : ahead;   // pages the stream will touch
    const bool populate = expect >= ((size_t)1 << 20);
    if (populate) t_ctx.pf.start(bytes, expect < ahead ? expect : ahead);
    do {
        // pixels in (the copy engine reads pageable memory at the link's rate on this platform, tools/ubench/host_copy.cpp),
        // kernels, then ONE read-back of length + liveness flag through pinned words, then exactly `len` bytes out
        if (hipMemcpyAsync(c->io_a.base, data, in_bytes, hipMemcpyHostToDevice, st) != hipSuccess) break;
        // A placement that waits on other sets (tree, look-back) bounds its spins; should that bound ever trip - never observed - the
        // image is encoded once more by the order-free form, in which no set waits for another.
        bool sound = false;
        for (int attempt = 0; attempt < 2 && !sound; ++attempt) {
            const int forced = c->enc_lookback;
            if (attempt) c->enc_lookback = 0;
            const int rc = qoimi_encode_batch(c, c->io_a.base, in_bytes, desc, 1, c->io_b.base, bound, (int*)c->io_c.base, st);
            c->enc_lookback = forced;
            if (rc != QOIMI_OK) break;
            if (hipMemcpyAsync(&c->host_word[4], c->io_c.base, sizeof(int), hipMemcpyDeviceToHost, st) != hipSuccess) break;
            if (hipMemcpyAsync(&c->host_word[5], c->last_enc_err, sizeof(uint32_t), hipMemcpyDeviceToHost, st) != hipSuccess) break;
            if (hipStreamSynchronize(st) != hipSuccess) break;
            sound = c->host_word[5] == 0;
        }
        const int len = (int)c->host_word[4];
        if (!sound || len < kHeaderBytes + kTrailerBytes || (size_t)len > bound) {
            t_error = "encode kernel reported a liveness failure";
            break;
        }
        if (populate) { t_ctx.pf.wait(); }
        if ((size_t)len > ahead) {                                      // longer than expected: an exact buffer instead
            free(bytes);
            bytes = (uint8_t*)malloc((size_t)len);
            if (!bytes) return NULL;
            prefault_pages(bytes, (size_t)len);
        }
        if (hipMemcpyAsync(bytes, c->io_b.base, (size_t)len, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) break;
        c->last_drop_len = (size_t)len;
        *out_len = len;
        result = bytes;
    } while (0);
    if (!result) { if (populate) t_ctx.pf.wait(); free(bytes); }
    return result;
}

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

extern "C" void* qoi_decode(const void* data, int size, qoi_desc* desc, int channels) {
    if (!data || !desc || (channels != 0 && channels != 3 && channels != 4) ||
        size < kHeaderBytes + kTrailerBytes) return NULL;                 // qoi.h:497-503
    const uint8_t* bytes = (const uint8_t*)data;
    const bool magic_ok = memcmp(bytes, "qoif", 4) == 0;
    desc->width = be32(bytes + 4);                                        // filled before validation, qoi.h:507-511
    desc->height = be32(bytes + 8);
    desc->channels = bytes[12];
    desc->colorspace = bytes[13];
    if (!desc_ok(desc) || !magic_ok) return NULL;                         // qoi.h:513-521
    const int och = channels ? channels : desc->channels;                 // qoi.h:523-525
    const size_t out_bytes = (size_t)desc->width * desc->height * (size_t)och;

    qoimi_ctx* c = thread_ctx();
    if (!c) return NULL;
    DeviceGuard guard(c->device);
    hipStream_t st = c->own_stream;
    if (c->io_a.reserve((size_t)size + 16) || c->io_b.reserve(out_bytes + 16)) return NULL;
    uint8_t* pixels = (uint8_t*)malloc(out_bytes);                        // qoi.h:527-531
    if (!pixels) return NULL;
    // The pages of the result are populated by the calling thread's two parked helpers while the stream goes in and the
    // kernels run (33 MB take one thread ~1.2 ms - the kernel zeroes them); the two copies alone take 0.79 ms (bench.py
    // dropin_host_pointers).  (Copying back in 4 MiB parts behind the populating threads instead of after them: 4.2 ms -
    // every pageable copy pins its pages under the same lock the populating threads hold.)
    const bool populate = out_bytes >= ((size_t)4 << 20);
    if (populate) t_ctx.pf.start(pixels, out_bytes);
    bool ok = hipMemcpyAsync(c->io_a.base, data, (size_t)size, hipMemcpyHostToDevice, st) == hipSuccess &&
              qoimi_decode_batch(c, c->io_a.base, (size_t)size, &size, desc, 1, channels, c->io_b.base, out_bytes, st) == QOIMI_OK;
    if (populate) t_ctx.pf.wait();
    ok = ok && hipMemcpyAsync(pixels, c->io_b.base, out_bytes, hipMemcpyDeviceToHost, st) == hipSuccess &&
         hipStreamSynchronize(st) == hipSuccess;
    if (!ok) { free(pixels); return NULL; }
    return pixels;
}

// stdio wrappers, same observable behaviour as qoi.h:595-646.  Like the reference (qoi.h:51-58,592: QOI_NO_STDIO), a build with
// -DQOI_NO_STDIO leaves them out (make -C qoi_amd/csrc NO_STDIO=1 -> libqoi_mi355x_nostdio.so).
#ifndef QOI_NO_STDIO
extern "C" int qoi_write(const char* filename, const void* data, const qoi_desc* desc) {
    FILE* f = fopen(filename, "wb");
    if (!f) return 0;
    int size = 0;
    void* encoded = qoi_encode(data, desc, &size);
    if (!encoded) { fclose(f); return 0; }
    fwrite(encoded, 1, (size_t)size, f);
    fflush(f);
    const int err = ferror(f);
    fclose(f);
    free(encoded);
    return err ? 0 : size;
}

extern "C" void* qoi_read(const char* filename, qoi_desc* desc, int channels) {
    FILE* f = fopen(filename, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    const long size = ftell(f);
    if (size <= 0 || size > 0x7FFFFFFFL || fseek(f, 0, SEEK_SET) != 0) { fclose(f); return NULL; }
    void* data = malloc((size_t)size);
    if (!data) { fclose(f); return NULL; }
    const size_t got = fread(data, 1, (size_t)size, f);
    fclose(f);
    void* pixels = (got != (size_t)size) ? NULL : qoi_decode(data, (int)got, desc, channels);
    free(data);
    return pixels;
}
#endif  // QOI_NO_STDIO
