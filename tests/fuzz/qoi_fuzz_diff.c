/* qoi_fuzz_diff.c — coverage-guided differential fuzzing of the drop-in qoi_decode (TEST ONLY).
 *
 * Same shape and input convention as the reference's harness (qoifuzz.c:20-32: the first four bytes are the `channels`
 * argument, the rest is the stream), built the same way (libFuzzer + AddressSanitizer; UBSan on top) - but where qoifuzz.c
 * only asks "does it crash", this one runs the UNMODIFIED reference decoder (oracle/_ref/libqoiref.so, symbols renamed to
 * ref_qoi_*) on the same bytes in the same process and compares: NULL-ness, the qoi_desc both wrote, every pixel byte.
 * The host shim of libqoi_mi355x (argument checks, header parsing, arena carving, stride arithmetic, helper thread) is
 * compiled with the sanitizers and libFuzzer's coverage instrumentation (tests/fuzz/Makefile: libqoi_mi355x_asan.so);
 * the gfx950 kernels are not instrumented - their errors show up as differences.
 *
 * Streams whose (valid) header asks for more than 2^20 pixels are skipped: both decoders would fill hundreds of megabytes
 * per input.  Headers the reference rejects are NOT skipped - both sides must return NULL and leave the same desc. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { unsigned int width, height; unsigned char channels, colorspace; } qoi_desc;   /* qoi.h:236-241 */
void *qoi_decode(const void *data, int size, qoi_desc *desc, int channels);                   /* libqoi_mi355x (qoi.h:289) */
void *ref_qoi_decode(const void *data, int size, qoi_desc *desc, int channels);               /* the reference itself */

static unsigned long long n_inputs, n_decoded, n_null, n_skipped;

static uint32_t be32(const uint8_t *p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static void report(void) {
    fprintf(stderr, "qoi_fuzz_diff: %llu inputs, %llu decoded by both (pixels identical), %llu rejected by both, %llu skipped (> 2^20 pixels)\n",
            n_inputs, n_decoded, n_null, n_skipped);
}

int LLVMFuzzerTestOneInput(const uint8_t *data, size_t size) {
    static int first = 1;
    if (first) { first = 0; atexit(report); }
    if (size < 4) return 0;                                                /* qoifuzz.c:21-23 */
    ++n_inputs;
    int channels;
    memcpy(&channels, data, 4);                                            /* qoifuzz.c:24-25 */
    const uint8_t *s = data + 4;
    const int n = (int)(size - 4);
    if (n >= 14 && memcmp(s, "qoif", 4) == 0) {
        const uint64_t w = be32(s + 4), h = be32(s + 8);
        if (w && h && s[12] >= 3 && s[12] <= 4 && s[13] <= 1 && h < 400000000u / w && w * h > (1u << 20)) { ++n_skipped; return 0; }
    }
    qoi_desc da, db;
    memset(&da, 0xA5, sizeof da); memset(&db, 0xA5, sizeof db);
    void *a = qoi_decode(s, n, &da, channels);
    void *b = ref_qoi_decode(s, n, &db, channels);
    if ((a == NULL) != (b == NULL)) {
        fprintf(stderr, "MISMATCH: ours %s, reference %s (size %d, channels arg %d)\n", a ? "decoded" : "NULL", b ? "decoded" : "NULL", n, channels);
        abort();
    }
    /* the fields are compared one by one: padding bytes of the struct are nobody's business */
    if (da.width != db.width || da.height != db.height || da.channels != db.channels || da.colorspace != db.colorspace) {
        fprintf(stderr, "MISMATCH: desc ours %u x %u c%u s%u, reference %u x %u c%u s%u\n", da.width, da.height, da.channels, da.colorspace,
                db.width, db.height, db.channels, db.colorspace);
        abort();
    }
    if (a) {
        const size_t och = channels ? (size_t)channels : da.channels;
        const size_t bytes = (size_t)da.width * da.height * och;
        if (memcmp(a, b, bytes) != 0) {
            size_t i = 0;
            while (((const uint8_t *)a)[i] == ((const uint8_t *)b)[i]) ++i;
            fprintf(stderr, "MISMATCH: pixel byte %zu of %zu: ours %u, reference %u (%u x %u, channels arg %d)\n", i, bytes,
                    ((const uint8_t *)a)[i], ((const uint8_t *)b)[i], da.width, da.height, channels);
            abort();
        }
        ++n_decoded;
    } else ++n_null;
    free(a); free(b);                                                      /* qoifuzz.c:28-30: plain malloc memory on both sides */
    return 0;
}
