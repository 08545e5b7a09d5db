"""Host rehearsal of the encoder's matrix-pipe classification (qoi_amd/csrc/qoi_encode.hip, CLS 1).

The kernel hands the (previous pixel, pixel) register pair of every lane - the previous pixel's alpha byte overwritten with the
constant 0xFF - to one v_mfma_i32_32x32x16_i8 whose A operand holds a block-diagonal coefficient matrix; every lane gets back
sixteen integer-linear forms of its own eight bytes and builds the literal chunk word of qoi.h:438-474 from their low bytes.
This module restates

  * the instruction as the ISA defines it (signed 8-bit inputs, 32-bit accumulation; operand / result layout: lane l holds
    row or column l % 32 and the K range 8 (l / 32) .. + 7 of A / B, and rows 8 q + 4 (l / 32) + r, q, r = 0..3, of column
    l % 32 of D in register 4 q + r),
  * mat_const_init / mat_classify / literal_word_mat of the kernel, line by line,

so that tests/test_encode_matclass.py can compare the words a wavefront would produce with a direct evaluation of the
reference's rules on the CPU.  Nothing here is used by the product path.
"""
import numpy as np


def coef4(c0, c1, c2, c3):
    return np.array([c0, c1, c2, c3], dtype=np.int64)


def coef_row(pr, pg, pb, bias, r, g, b, a):
    """8 coefficients over (previous pixel r g b, the constant -1, pixel r g b a) - byte order of the register pair as loaded;
    the constant term rides on the alpha byte of the previous pixel, which the kernel overwrites with 0xFF (= -1)"""
    row = np.concatenate([coef4(pr, pg, pb, -bias), coef4(r, g, b, a)])
    assert row.min() >= -128 and row.max() <= 127, row
    return row.astype(np.int8)


# form number -> row (mat_const_init of the kernel); forms 6, 7, 12..15 carry nothing
FORMS = {
    0: coef_row(-1, 0, 0, 2, 1, 0, 0, 0),                 # vr + 2
    1: coef_row(0, -1, 0, 2, 0, 1, 0, 0),                 # vg + 2
    2: coef_row(0, 0, -1, 2, 0, 0, 1, 0),                 # vb + 2
    3: coef_row(0, 0, 0, 0, 12, 20, 28, 44),              # 4 * hash
    4: coef_row(-1, 1, 0, 8, 1, -1, 0, 0),                # vr - vg + 8
    5: coef_row(0, 1, -1, 8, 0, -1, 1, 0),                # vb - vg + 8
    8: coef_row(0, -1, 0, 160 - 256, 0, 1, 0, 0),         # vg + 160 (mod 256)
    9: coef_row(-16, 17, -1, 136 - 256, 16, -17, 1, 0),   # 16 vr - 17 vg + vb + 136 (mod 256)
    10: coef_row(-16, -4, -1, 106, 16, 4, 1, 0),          # 16 vr + 4 vg + vb + 106
    11: coef_row(0, -1, 0, 32, 0, 1, 0, 0),               # vg + 32
}


def mat_const_init():
    """A operand of the 64 lanes: [64][8] int8 (mat_const_init of the kernel)"""
    a = np.zeros((64, 8), dtype=np.int8)
    for lane in range(64):
        row, half = lane & 31, lane >> 5
        if ((row >> 2) & 1) == half:
            f = 4 * (row >> 3) + (row & 3)
            if f in FORMS:
                a[lane] = FORMS[f]
    return a


def mfma_i32_32x32x16_i8(a_lanes, b_lanes):
    """D = A x B (C = 0).  a_lanes, b_lanes: [64][8] int8 (the eight bytes a lane holds).  Returns [64][16] wrapped to 32 bits.
    A[i][k]: lane i + 32 (k // 8), byte k % 8;  B[k][j]: lane j + 32 (k // 8), byte k % 8;
    D[i][j]: lane j + 32 ((i // 4) % 2), register 4 (i // 8) + i % 4."""
    A = np.zeros((32, 16), dtype=np.int64)
    B = np.zeros((16, 32), dtype=np.int64)
    for lane in range(64):
        h, t = lane >> 5, lane & 31
        A[t, 8 * h:8 * h + 8] = a_lanes[lane]
        B[8 * h:8 * h + 8, t] = b_lanes[lane]
    D = A @ B
    out = np.zeros((64, 16), dtype=np.int64)
    for lane in range(64):
        h, j = lane >> 5, lane & 31
        for v in range(16):
            out[lane, v] = D[8 * (v // 4) + 4 * h + (v % 4), j]
    return ((out + 2**31) % 2**32) - 2**31


def step_words(px, prev):
    """px, prev: [64] uint32 (little-endian r g b a).  Returns per lane (word, alpha_moved, hash4) as the kernel's
    mat_classify + literal_word_mat would: word = LUMA word (byte 0 in bits 0..7, byte 1 in bits 16..23, 0xFF on top), DIFF byte,
    or the long marker 0x40000000."""
    alpha_moved = (px >> 24) != (prev >> 24)                     # taken before the alpha byte is overwritten
    pm = (prev & 0x00FFFFFF) | 0xFF000000                        # v_perm_b32: r, g, b as they are, 0xFF on top
    b = np.zeros((64, 8), dtype=np.int8)
    b[:, 0:4] = pm.astype('<u4').view(np.uint8).reshape(64, 4).view(np.int8)
    b[:, 4:8] = px.astype('<u4').view(np.uint8).reshape(64, 4).view(np.int8)
    d = mfma_i32_32x32x16_i8(mat_const_init(), b).astype(np.int64) & 0xFFFFFFFF
    xr, xg, xb, h4, ur, ub = d[:, 0], d[:, 1], d[:, 2], d[:, 3], d[:, 4], d[:, 5]
    b0, b1, wd, ug = d[:, 8], d[:, 9], d[:, 10], d[:, 11]
    od = xr | xg | xb
    ol = ((ug >> 2) & 63) | ur | ub
    luma_ok = (ol & 0xFF) < 16
    diff_ok = (od & 0xFF) < 4
    luma_word = (b0 & 0xFF) | ((b1 & 0xFF) << 16) | 0xFF000000
    we = np.where(luma_ok, luma_word, 0x40000000)
    we = np.where(diff_ok, wd & 0xFF, we)
    we = np.where(alpha_moved, 0x40000000, we)
    return we.astype(np.uint32), alpha_moved, (h4 & 0xFC).astype(np.uint32)


def reference_words(px, prev):
    """The same from the reference's rules (qoi.h:322,438-474), one pixel at a time."""
    words, hashes = [], []
    for p, q in zip(px.tolist(), prev.tolist()):
        r, g, b_, a = p & 255, (p >> 8) & 255, (p >> 16) & 255, p >> 24
        pr, pg, pb, pa = q & 255, (q >> 8) & 255, (q >> 16) & 255, q >> 24
        sc = lambda v: ((v + 128) & 255) - 128          # signed char
        hashes.append(((r * 3 + g * 5 + b_ * 7 + a * 11) % 64) * 4)
        if a != pa:
            words.append(0x40000000); continue
        vr, vg, vb = sc(r - pr), sc(g - pg), sc(b_ - pb)
        vg_r, vg_b = sc(vr - vg), sc(vb - vg)
        if -3 < vr < 2 and -3 < vg < 2 and -3 < vb < 2:
            words.append(0x40 | (vr + 2) << 4 | (vg + 2) << 2 | (vb + 2))
        elif -9 < vg_r < 8 and -33 < vg < 32 and -9 < vg_b < 8:
            words.append((0x80 | (vg + 32)) | (((vg_r + 8) << 4 | (vg_b + 8)) << 16) | 0xFF000000)
        else:
            words.append(0x40000000)
    return np.array(words, dtype=np.uint32), np.array(hashes, dtype=np.uint32)
