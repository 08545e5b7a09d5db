/*
 * qoi_mi355x.h — C-ABI of libqoi_mi355x.so, the MI355X-native QOI encode/decode path.
 *
 * Part 1 is the DROP-IN boundary: the same four symbols, constants and struct the
 * reference exposes (phoboslab/qoi qoi.h:214-295).  A caller that includes the
 * reference's qoi.h WITHOUT defining QOI_IMPLEMENTATION (prototypes only) and links
 * this library gets the GPU path with no source change; see INTEGRATION.md.
 *
 * Part 2 is ADDITIVE (nothing like it exists in the reference): device-resident
 * batch entry points for callers whose pixels/streams already live in HBM.  Plain
 * pointers and sizes only — no torch / HIP types in any signature (a hipStream_t is
 * passed as void*).
 *
 * All work is done by hand-written gfx950 kernels (the .hip files under qoi_amd/csrc).  There is no
 * CPU fallback: without a usable GPU every entry point fails (NULL / negative status)
 * and qoimi_last_error() says why.
 */
#ifndef QOI_MI355X_H
#define QOI_MI355X_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------
 * Part 1 — drop-in for the reference API
 * ------------------------------------------------------------------------------------ */

#ifndef QOI_H            /* the reference header may already have declared these */
#define QOI_H

#define QOI_SRGB   0     /* qoi.h:233 */
#define QOI_LINEAR 1     /* qoi.h:234 */

/* qoi.h:236-241 — 12 bytes, fields at offsets 0/4/8/9. */
typedef struct {
    unsigned int width;
    unsigned int height;
    unsigned char channels;
    unsigned char colorspace;
} qoi_desc;

/* Replaces qoi.h:278 / implementation qoi.h:356-486.
 * Same contract: NULL on data/desc/out_len == NULL, width or height 0, channels not 3/4,
 * colorspace > 1 or height >= 400000000/width; otherwise a malloc()ed buffer (caller
 * free()s it) holding a stream BYTE-IDENTICAL to the reference encoder's, *out_len set.
 * The buffer holds at least *out_len bytes; the reference's is always the worst-case size
 * (qoi.h:374-379), which nothing in its contract lets a caller rely on. */
void *qoi_encode(const void *data, const qoi_desc *desc, int *out_len);

/* Replaces qoi.h:289 / implementation qoi.h:488-590.
 * Same contract incl. leniency (truncated streams repeat the last pixel, trailer content
 * ignored, over-long runs clipped, ...): result pixels are bit-identical to the
 * reference decoder's for EVERY input stream; desc is filled before validation. */
void *qoi_decode(const void *data, int size, qoi_desc *desc, int channels);

/* Replace qoi.h:252 / qoi.h:595-617 and qoi.h:265 / qoi.h:619-646 (stdio wrappers). */
int   qoi_write(const char *filename, const void *data, const qoi_desc *desc);
void *qoi_read(const char *filename, qoi_desc *desc, int channels);

#endif /* QOI_H */

/* ------------------------------------------------------------------------------------
 * Part 2 — additive device-resident API
 * ------------------------------------------------------------------------------------ */

typedef struct qoimi_ctx qoimi_ctx;

enum {
    QOIMI_OK            =  0,
    QOIMI_E_ARG         = -1,   /* argument rejected by the same rules as qoi.h:364-372 / 497-521 */
    QOIMI_E_NO_GPU      = -2,   /* no usable gfx950 device (none present, wrong architecture, driver missing) */
    QOIMI_E_NOMEM       = -3,
    QOIMI_E_INTERNAL    = -4    /* any other HIP runtime error (e.g. a rejected launch), or a device-side liveness bound tripped */
};

/* Synthetic content classes (qoi_amd/synth.py states the exact per-pixel function). */
enum { QOIMI_NOISE = 0, QOIMI_PHOTO = 1, QOIMI_UIFLAT = 2, QOIMI_CONSTANT = 3, QOIMI_PHOTO_HARD = 4, QOIMI_SPRITE_ALPHA = 5 };

/* Create / destroy a context bound to one GPU.  A context owns a growable device
 * workspace, so steady-state calls do no hipMalloc/hipFree.  ONE call at a time per
 * context: the workspace (scratch, flags, staging) belongs to the call in flight, so a
 * second qoimi_encode_batch / qoimi_decode_batch on the same context must not start -
 * on any stream - before the first one's work has completed (qoimi_decode_batch returns
 * complete; after qoimi_encode_batch synchronise the stream or call qoimi_encode_status).
 * Use one context per thread / per concurrent stream.  The drop-in functions above do
 * exactly that internally: every calling thread gets its own context and stream, so they
 * are re-entrant like the reference (qoi.h:339,357-362,489-495).  Entry points leave the
 * calling thread's current HIP device as they found it. */
int  qoimi_ctx_create(int device, qoimi_ctx **out);
void qoimi_ctx_destroy(qoimi_ctx *ctx);

/* Thread-local description of the last failure in this thread ("" if none). */
const char *qoimi_last_error(void);

/* Worst-case stream size w*h*(channels+1)+14+8 — the reference's allocation, qoi.h:374-376.
 * Returns 0 if desc is rejected. */
size_t qoimi_encode_bound(const qoi_desc *desc);

/* Encode n_images equally-shaped images that live in device memory.
 *   d_pixels       device pointer; image i starts at d_pixels + i*pixel_stride (tightly
 *                  packed row-major w*h*channels bytes, as qoi.h:406-413 reads them)
 *   d_streams      device pointer; stream i is written at d_streams + i*stream_stride,
 *                  stream_stride >= qoimi_encode_bound(desc)
 *   d_stream_len   device int[n_images]; receives each stream's length (= *out_len)
 *   stream         hipStream_t (as void*), NULL = default stream
 * Asynchronous: kernels are enqueued on `stream`; nothing is synchronised. */
int qoimi_encode_batch(qoimi_ctx *ctx, const void *d_pixels, size_t pixel_stride,
                       const qoi_desc *desc, int n_images,
                       void *d_streams, size_t stream_stride, int *d_stream_len,
                       void *stream);

/* The same for images of DIFFERENT shapes and channel counts in one call (a directory of images, qoibench.c:491-555):
 *   pixel_offsets  HOST size_t[n_images]: image i starts at d_pixels + pixel_offsets[i] (tightly packed, as above)
 *   descs          HOST qoi_desc[n_images]: every one must pass the rules of qoi.h:364-372
 *   stream_offsets HOST size_t[n_images]: stream i is written at d_streams + stream_offsets[i]; the caller leaves
 *                  qoimi_encode_bound(&descs[i]) bytes there
 * Streams are byte-identical to the reference encoder's.  Every set of slabs parks its bytes in a scratch slot of its own and two
 * more passes place them (no set waits for another, whatever the mix of sizes): about 5 bytes of workspace per pixel of the call.
 * Enqueues on `stream`; the call itself waits for the stream's earlier work once (its image table travels through pinned staging). */
int qoimi_encode_images(qoimi_ctx *ctx, const void *d_pixels, const size_t *pixel_offsets, const qoi_desc *descs, int n_images,
                        void *d_streams, const size_t *stream_offsets, int *d_stream_len, void *stream);

/* The encoder's colour-table probe uses one LDS exchange instruction per 64 pixels and relies on the LDS serving the lanes of
 * that instruction in ascending order - a MEASURED property of gfx950, not a documented one (qoi_amd/csrc/qoi_encode.hip).  It
 * is measured alone and under contention when a context is created (a failure selects the order-independent probe) and again
 * every 256 encode calls of the context (env QOIMI_ENC_RECHECK_EVERY) on the context's private stream; that repeat is looked at
 * by the NEXT encode call.  Failure latency, stated plainly: if a repeat ever failed, up to 2 x 256 earlier calls of the
 * context would already have returned streams made with the suspect probe.  The context then switches to the order-independent
 * probe for good; the call that notices is encoded with it and returns QOIMI_OK (its own stream is sound), qoimi_last_error()
 * says what happened, the next qoimi_encode_status() returns QOIMI_E_INTERNAL ONCE, and this counter returns how many calls were
 * made since the launch of the last check that passed (the calls before the failed repeat was launched and the ones made while
 * it ran) - re-verify or re-encode those.  0 = never.
 * QOIMI_ENC_PROBE=0 in the environment selects the order-independent probe from the start (about 1.5 x the encode time). */
long long qoimi_encode_suspect_calls(qoimi_ctx *ctx);

/* Synchronise `stream`.  If a placement wait of the last qoimi_encode_batch on this context gave up (never observed: a set waits for
 * sets that are resident or done - with tickets by construction, for calls of fewer than 8 images by the order the dispatcher starts
 * workgroups in, which launches of other streams could in principle disturb; such waits are bounded and end each other) the call is
 * encoded again here, order-free, from the caller's buffers - which the caller must therefore not have released or overwritten before
 * asking.  QOIMI_E_INTERNAL if that failed too, or - once - after a repeat of the LDS-order self-test failed (see
 * qoimi_encode_suspect_calls); QOIMI_OK otherwise.  A caller that only synchronises its stream never learns of a wait that gave up:
 * call this before reading the streams. */
int qoimi_encode_status(qoimi_ctx *ctx, void *stream);

/* How qoimi_encode_batch hands out the work units of calls of fewer than 8 images (tree placement, where a set waits for the byte
 * counts of lower-numbered sets).  by_workgroup_index = 0 (default): by one ticket per workgroup, i.e. in START order - whatever a set
 * waits for is running or done, on a shared device too; a caller that only synchronises its stream reads complete streams.
 * by_workgroup_index = 1: by workgroup index - 4 us less per 4K frame, correct only as long as the dispatcher starts lower-numbered
 * workgroups no later than higher ones (true of one launch on an idle device); its waits are bounded, a wait that gives up ends the
 * launch's waits and leaves an error flag, and the caller MUST call qoimi_encode_status before reading the streams (it encodes the
 * call again order-free).  The drop-in qoi_encode uses this form and does that by itself.  Returns QOIMI_OK or QOIMI_E_ARG. */
int qoimi_set_encode_small_call_order(qoimi_ctx *ctx, int by_workgroup_index);

/* Calls of this context that qoimi_encode_status encoded again order-free because a placement wait had given up (0: never). */
long long qoimi_encode_retries(qoimi_ctx *ctx);

/* Decode n_images streams that live in device memory.
 *   d_streams      stream i starts at d_streams + i*stream_stride and is sizes[i] bytes
 *   sizes          HOST int[n_images] (the `size` argument of qoi.h:289 per image)
 *   descs          HOST qoi_desc[n_images]: the header of each stream as parsed by the
 *                  caller (what qoi_decode would write to *desc); the images may differ in
 *                  shape - each must decode to w*h*out_channels <= pixel_stride bytes - but
 *                  share the output channel count (channels, or their headers' when it is 0)
 *   channels       0, 3 or 4 — as qoi.h:289
 *   d_pixels       image i is written at d_pixels + i*pixel_stride
 * Synchronous: returns when every pixel is written and verified (the exactness check of the speculative decoder is read
 * before returning).  For calls of a few images the call returns on result words its last launch writes to pinned memory
 * when everything in front of it is done; that launch itself - it writes nothing a caller can see - may still be retiring on
 * `stream` for a few microseconds.  Work the caller enqueues on `stream` is ordered behind it; the context's next call waits
 * for it by itself if it comes on another stream. */
int qoimi_decode_batch(qoimi_ctx *ctx, const void *d_streams, size_t stream_stride,
                       const int *sizes, const qoi_desc *descs, int n_images, int channels,
                       void *d_pixels, size_t pixel_stride, void *stream);

/* Fill device memory with synthetic RGBA frames frame_id = first_frame .. first_frame+n-1
 * (benchmark/test utility; same function of (kind, seed, frame, pixel) as synth.py). */
int qoimi_synth_frames(qoimi_ctx *ctx, int kind, unsigned seed, unsigned first_frame,
                       int n_frames, unsigned width, unsigned height,
                       void *d_pixels, size_t pixel_stride, void *stream);

/* 64-bit content hash of n_streams streams in device memory (stream i: d_stream_len[i] bytes at d_streams + i*stream_stride),
 * written to the device array d_hash[n_streams]; asynchronous on `stream`.  Benchmark / test utility: whole batches are compared
 * with hashes of the REFERENCE encoder's streams without copying the streams out (bench.py, qoi_amd/synth.py: stream_hash64 states
 * the function: sum over the 8-byte little-endian words w_j, last one zero-padded, of splitmix64(w_j + (j+1) * 0x9E3779B97F4A7C15)). */
int qoimi_hash_streams(qoimi_ctx *ctx, const void *d_streams, size_t stream_stride, const int *d_stream_len, int n_streams,
                       unsigned long long *d_hash, void *stream);

/* Device memory the context's growable arenas hold at the moment (bytes): [0] encode workspace, [1] decode workspace,
 * [2] staging buffers of the host-pointer entry points (qoi_encode / qoi_decode of the calling thread's context). */
void qoimi_workspace_bytes(qoimi_ctx *ctx, size_t out[3]);

/* Caps the chunk-record arena of this context's decode calls (bytes; at least 1 MiB): a call whose streams need more - four bytes
 * per stream byte, the worst case - is decoded as consecutive sub-batches of whole images through the same arena, and `release`
 * != 0 gives the arena grown so far back to the device.  Default: a sixth of the device's memory, 48 GiB at most (1024 4K
 * photographs in one piece: 45 GB of workspace); 24 GiB decodes them as two sub-batches in 27 GB at +1 % of the time
 * (DESIGN.md section 4).  The pixels are the same either way.  Returns QOIMI_OK or QOIMI_E_ARG. */
int qoimi_set_decode_record_cap(qoimi_ctx *ctx, size_t bytes, int release);

/* Counters of the last decode on this context: [0] speculation rounds, [1] segments
 * re-decoded after a failed check, [2] total segments, [3] segments whose entry position
 * needed the full five-phase parse (look-back synchronisation did not settle). */
void qoimi_decode_stats(qoimi_ctx *ctx, long long out[4]);

/* Per-kernel timing with HIP events recorded on the launch stream (what bench.py's roofline
 * figure is computed from).  qoimi_set_profiling(ctx,1) enables it and resets the
 * accumulators; qoimi_get_profile synchronises `stream` and returns, per kernel index,
 * accumulated milliseconds and launch counts (arrays of `cap` entries; return value =
 * number of kernels); qoimi_kernel_name(i) names index i. */
int qoimi_set_profiling(qoimi_ctx *ctx, int on);
int qoimi_get_profile(qoimi_ctx *ctx, void *stream, double *ms, long long *calls, int cap);
const char *qoimi_kernel_name(int index);

/* Library/version string, e.g. "qoi_mi355x 0.1 gfx950". */
const char *qoimi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* QOI_MI355X_H */
