"""Host rehearsal of the encoder's matrix-pipe classification (qoi_amd/csrc/qoi_encode.hip, CLS 1).

The kernel hands the (previous pixel, pixel) register pair of every lane to three v_mfma_i32_16x16x32_i8 instructions whose
A operands hold block-diagonal coefficient matrices; every lane gets back twelve integer-linear forms of its own eight bytes
and builds the literal chunk word of qoi.h:438-474 from their low bytes.  This module restates

  * the instruction as the ISA defines it (signed 8-bit inputs, 32-bit accumulation; operand / result layout: lane l holds
    row or column l % 16 and the K range 8 (l / 16) .. + 7 of A / B, and rows 4 (l / 16) + r, r = 0..3, of column l % 16 of D),
  * mat_const_init / mat_classify / literal_word_mat of the kernel, line by line,

so that tests/test_encode_matclass.py can compare the words a wavefront would produce with a direct evaluation of the
reference's rules on the CPU.  Nothing here is used by the product path.
"""
import numpy as np


def coef4(c0, c1, c2, c3):
    return np.array([c0, c1, c2, c3], dtype=np.int8)


def coef_row(on_prev, on_px):
    """8 coefficients over (previous pixel r g b a, pixel r g b a) - byte order of the register pair as loaded"""
    return np.concatenate([on_prev, on_px]).astype(np.int8)


ZERO4 = coef4(0, 0, 0, 0)
DR, DG, DB = coef4(1, 0, 0, 0), coef4(0, 1, 0, 0), coef4(0, 0, 1, 0)
NDR, NDG, NDB = coef4(-1, 0, 0, 0), coef4(0, -1, 0, 0), coef4(0, 0, -1, 0)

M1_ROWS = [coef_row(NDR, DR), coef_row(NDG, DG), coef_row(NDB, DB), coef_row(ZERO4, coef4(12, 20, 28, 44))]
M2_ROWS = [coef_row(coef4(-1, 1, 0, 0), coef4(1, -1, 0, 0)), coef_row(coef4(0, 1, -1, 0), coef4(0, -1, 1, 0)),
           coef_row(ZERO4, ZERO4), coef_row(ZERO4, ZERO4)]
M3_ROWS = [coef_row(NDG, DG), coef_row(coef4(-16, 17, -1, 0), coef4(16, -17, 1, 0)),
           coef_row(coef4(-16, -4, -1, 0), coef4(16, 4, 1, 0)), coef_row(NDG, DG)]
C1 = np.array([2, 2, 2, 2], dtype=np.int64)
C2 = np.array([8, 8, 8, 8], dtype=np.int64)
C3 = np.array([160, 136, 106, 32], dtype=np.int64)


def mat_const_init(rows):
    """A operand of the 64 lanes: [64][8] int8 (mat_const_init of the kernel)"""
    a = np.zeros((64, 8), dtype=np.int8)
    for lane in range(64):
        row, grp = lane & 15, lane >> 4
        if (row >> 2) == grp:
            a[lane] = rows[row & 3]
    return a


def mfma_i32_16x16x32_i8(a_lanes, b_lanes, c_lanes):
    """D = A x B + C.  a_lanes, b_lanes: [64][8] int8 (the eight bytes a lane holds), c_lanes: [64][4].  Returns [64][4] int64
    wrapped to 32 bits.  A[i][k]: lane i + 16 (k // 8), byte k % 8;  B[k][j]: lane j + 16 (k // 8), byte k % 8;
    D[i][j]: lane j + 16 (i // 4), register i % 4."""
    A = np.zeros((16, 32), dtype=np.int64)
    B = np.zeros((32, 16), dtype=np.int64)
    for lane in range(64):
        g, t = lane >> 4, lane & 15
        A[t, 8 * g:8 * g + 8] = a_lanes[lane]
        B[8 * g:8 * g + 8, t] = b_lanes[lane]
    D = A @ B
    out = np.zeros((64, 4), dtype=np.int64)
    for lane in range(64):
        g, j = lane >> 4, lane & 15
        for r in range(4):
            out[lane, r] = D[4 * g + r, j] + c_lanes[lane, r]
    return ((out + 2**31) % 2**32) - 2**31


def step_words(px, prev):
    """px, prev: [64] uint32 (little-endian r g b a).  Returns per lane (word, alpha_moved, hash4) as the kernel's
    literal_word_mat would: word = LUMA word (byte 0 in bits 0..7, byte 1 in bits 16..23, 0xFF on top), DIFF byte, or the long
    marker 0x40000000."""
    b = np.zeros((64, 8), dtype=np.int8)
    b[:, 0:4] = prev.astype('<u4').view(np.uint8).reshape(64, 4).view(np.int8)
    b[:, 4:8] = px.astype('<u4').view(np.uint8).reshape(64, 4).view(np.int8)
    d1 = mfma_i32_16x16x32_i8(mat_const_init(M1_ROWS), b, np.tile(C1, (64, 1)))
    d2 = mfma_i32_16x16x32_i8(mat_const_init(M2_ROWS), b, np.tile(C2, (64, 1)))
    d3 = mfma_i32_16x16x32_i8(mat_const_init(M3_ROWS), b, np.tile(C3, (64, 1)))
    u = lambda v: v.astype(np.int64) & 0xFFFFFFFF
    xr, xg, xb, h4 = u(d1[:, 0]), u(d1[:, 1]), u(d1[:, 2]), u(d1[:, 3])
    ur, ub = u(d2[:, 0]), u(d2[:, 1])
    b0, b1, wd, ug = u(d3[:, 0]), u(d3[:, 1]), u(d3[:, 2]), u(d3[:, 3])
    od = xr | xg | xb
    ol = ((ug >> 2) & 63) | ur | ub
    luma_ok = (ol & 0xFF) < 16
    diff_ok = (od & 0xFF) < 4
    alpha_moved = (px >> 24) != (prev >> 24)
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
