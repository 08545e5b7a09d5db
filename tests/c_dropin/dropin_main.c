/* dropin_main.c — a C caller of the reference's public API, compiled against the reference's own header
 * WITHOUT QOI_IMPLEMENTATION (prototypes only, qoi.h:214-295) and linked with libqoi_mi355x.so: the link-time swap
 * INTEGRATION.md section 1 describes, exercised for real.  TEST ONLY.
 *
 *   dropin roundtrip W H C     encode + decode + memcmp (qoibench.c:408-417), then qoi_write + qoi_read + memcmp
 *                              (the call pattern of qoiconv.c:60,76); prints "ok <stream bytes> <crc32 of the stream>"
 *   dropin fuzz FILE...        the body of qoifuzz.c:20-32 per file: first four bytes = channels argument, the rest a stream;
 *                              prints one line per file: "null" or "<w> <h> <channels> <colorspace> <crc32 of the pixels>"
 *                              (many files per process: a process start costs a HIP initialisation, seconds on some boxes)
 *   dropin header              which header this binary was compiled against
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include QOI_HEADER_FILE                      /* "qoi.h" of the reference, or include/qoi_mi355x.h where that tree is absent */

#ifdef QOI_IMPLEMENTATION
#error "the drop-in test must not compile the reference implementation"
#endif

static unsigned crc32_of(const unsigned char *p, size_t n) {
    unsigned c = 0xFFFFFFFFu;
    for (size_t i = 0; i < n; ++i) {
        c ^= p[i];
        for (int k = 0; k < 8; ++k) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1u)));
    }
    return ~c;
}

static int roundtrip(unsigned w, unsigned h, int ch) {
    size_t n = (size_t)w * h * (size_t)ch;
    unsigned char *px = (unsigned char *)malloc(n);
    unsigned s = 12345u;
    for (size_t i = 0; i < (size_t)w * h; ++i) {          /* ramps + a little noise + some flat stretches */
        s = s * 1664525u + 1013904223u;
        unsigned x = (unsigned)(i % w), y = (unsigned)(i / w);
        unsigned flat = ((x / 37u + y / 11u) % 5u) == 0u;
        px[i * ch + 0] = (unsigned char)(flat ? 200 : x / 3u + ((s >> 24) & 3u));
        px[i * ch + 1] = (unsigned char)(flat ? 100 : y / 2u + ((s >> 20) & 3u));
        px[i * ch + 2] = (unsigned char)(flat ? 50 : (x + y) / 5u + ((s >> 16) & 1u));
        if (ch == 4) px[i * ch + 3] = (unsigned char)(((x / 64u) & 1u) ? 255 : 128 + ((s >> 12) & 1u));
    }
    qoi_desc d;
    d.width = w; d.height = h; d.channels = (unsigned char)ch; d.colorspace = QOI_SRGB;
    int len = 0;
    void *enc = qoi_encode(px, &d, &len);
    if (!enc) { printf("encode failed\n"); return 1; }
    qoi_desc dd;
    void *dec = qoi_decode(enc, len, &dd, ch);
    if (!dec || dd.width != w || dd.height != h || dd.channels != ch || memcmp(dec, px, n) != 0) { printf("decode mismatch\n"); return 1; }
    free(dec);
    const char *path = "/tmp/qoi_mi355x_dropin_test.qoi";
    int written = qoi_write(path, px, &d);
    if (written != len) { printf("qoi_write returned %d, expected %d\n", written, len); return 1; }
    qoi_desc rd;
    void *back = qoi_read(path, &rd, 0);
    if (!back || rd.width != w || rd.height != h || rd.channels != ch || memcmp(back, px, n) != 0) { printf("qoi_read mismatch\n"); return 1; }
    free(back);
    remove(path);
    printf("ok %d %08x\n", len, crc32_of((const unsigned char *)enc, (size_t)len));
    free(enc);
    free(px);
    return 0;
}

static int fuzz(const char *file) {
    FILE *f = fopen(file, "rb");
    if (!f) return 2;
    fseek(f, 0, SEEK_END);
    long size = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char *data = (unsigned char *)malloc((size_t)size + 8);
    if (fread(data, 1, (size_t)size, f) != (size_t)size) { fclose(f); return 2; }
    fclose(f);
    if (size < 4) { printf("null\n"); return 0; }
    int channels;
    memcpy(&channels, data, 4);                            /* qoifuzz.c:24-27 */
    qoi_desc desc;
    void *decoded = qoi_decode(data + 4, (int)(size - 4), &desc, channels);
    if (decoded != NULL) {
        size_t och = channels ? (size_t)channels : desc.channels;
        printf("%u %u %u %u %08x\n", desc.width, desc.height, (unsigned)desc.channels, (unsigned)desc.colorspace,
               crc32_of((const unsigned char *)decoded, (size_t)desc.width * desc.height * och));
        free(decoded);
    } else {
        printf("null\n");
    }
    free(data);
    return 0;
}

int main(int argc, char **argv) {
    if (argc >= 2 && strcmp(argv[1], "header") == 0) { printf("%s\n", QOI_HEADER_NAME); return 0; }
    if (argc >= 5 && strcmp(argv[1], "roundtrip") == 0) return roundtrip((unsigned)atoi(argv[2]), (unsigned)atoi(argv[3]), atoi(argv[4]));
    if (argc >= 3 && strcmp(argv[1], "fuzz") == 0) {
        for (int i = 2; i < argc; ++i) { const int rc = fuzz(argv[i]); if (rc) return rc; }
        return 0;
    }
    fprintf(stderr, "usage: dropin roundtrip W H C | fuzz FILE... | header\n");
    return 2;
}
