/* PNG scanline reconstruction (ISO/IEC 15948 section 9.2) for tools/png_io.py - the Average and Paeth
 * filters depend on the reconstructed left neighbour, which is slow in numpy.  Built on first use with gcc. */
#include <stdint.h>
#include <stdlib.h>

/* raw: h * (1 + stride) filtered bytes; out: h * stride; returns 0, or -1 on a bad filter type */
int png_unfilter(const uint8_t *raw, uint8_t *out, int h, int stride, int bpp) {
    for (int y = 0; y < h; y++) {
        const uint8_t *in = raw + (size_t)y * (stride + 1);
        const int ft = in[0];
        const uint8_t *line = in + 1;
        uint8_t *cur = out + (size_t)y * stride;
        const uint8_t *prior = y ? cur - stride : 0;
        if (ft > 4) return -1;
        for (int x = 0; x < stride; x++) {
            const int a = x >= bpp ? cur[x - bpp] : 0;
            const int b = prior ? prior[x] : 0;
            const int c = (prior && x >= bpp) ? prior[x - bpp] : 0;
            int pred = 0;
            if (ft == 1) pred = a;
            else if (ft == 2) pred = b;
            else if (ft == 3) pred = (a + b) >> 1;
            else if (ft == 4) {
                const int p = a + b - c, pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
                pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
            }
            cur[x] = (uint8_t)(line[x] + pred);
        }
    }
    return 0;
}
