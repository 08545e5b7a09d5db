/*
 * qoi_oracle.h — C-ABI of the CPU oracle (TEST INFRASTRUCTURE ONLY; see qoi_oracle.c).
 * Mirrors the reference entry points qoi.h:278 (qoi_encode) and qoi.h:289 (qoi_decode)
 * under an oracle_ prefix so it can share a process with the product library
 * and with oracle/_ref/libqoiref.so.
 */
#ifndef QOI_ORACLE_H
#define QOI_ORACLE_H
#ifdef __cplusplus
extern "C" {
#endif

/* Same layout as the reference's qoi_desc (qoi.h:236-241): 12 bytes, fields at 0/4/8/9. */
typedef struct {
    unsigned int width;
    unsigned int height;
    unsigned char channels;
    unsigned char colorspace;
} oracle_qoi_desc;

void *oracle_qoi_encode(const void *data, const oracle_qoi_desc *desc, int *out_len);
void *oracle_qoi_decode(const void *data, int size, oracle_qoi_desc *desc, int channels);
void  oracle_qoi_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
