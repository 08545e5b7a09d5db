"""numpy model of the slab-parallel exact encoder (design check for csrc/qoi_encode.hip).

Every output byte of the reference encoder is a pure function of the input pixels
(SURVEY.md Appendix C.1).  This model states that function the way the GPU kernels
evaluate it — per-pixel rules + three carried quantities per slab (colour table,
last-edge position, byte offset) — and is compared with the reference in
tests/test_encode_model.py.  It is a design aid, not product code.
"""
import numpy as np

INIT_PREV = np.uint32(0xFF000000)  # {0,0,0,255} as little-endian r,g,b,a


def hash_slots(px):
    r = px & 0xFF; g = (px >> 8) & 0xFF; b = (px >> 16) & 0xFF; a = (px >> 24) & 0xFF
    return ((r * 3 + g * 5 + b * 7 + a * 11) & 63).astype(np.int64)


def slab_summaries(px, slab):
    """Pass E1: per slab, last edge value per slot (+valid) and last edge position (-1: none)."""
    n = len(px)
    prev = np.concatenate([[INIT_PREV], px[:-1]])
    edge = px != prev
    h = hash_slots(px)
    ns = (n + slab - 1) // slab
    tab = np.zeros((ns, 64), dtype=np.uint32); valid = np.zeros((ns, 64), dtype=bool)
    le = np.full(ns, -1, dtype=np.int64)
    for s in range(ns):
        lo, hi = s * slab, min(n, (s + 1) * slab)
        idx = np.nonzero(edge[lo:hi])[0] + lo
        if len(idx):
            le[s] = idx[-1]
            tab[s, h[idx]] = px[idx]          # later assignments win -> last edge per slot
            valid[s, h[idx]] = True
    return tab, valid, le


def scan_entries(tab, valid, le):
    """Pass E2: exclusive 'latest valid per slot' / max scans over slabs."""
    ns = len(le)
    etab = np.zeros_like(tab); ele = np.full(ns, -1, dtype=np.int64)
    cur = np.zeros(64, dtype=np.uint32); curle = -1
    for s in range(ns):
        etab[s] = cur; ele[s] = curle
        cur = np.where(valid[s], tab[s], cur)
        curle = max(curle, le[s])
    return etab, ele


def encode_slab(px, lo, hi, n, entry_tab, entry_le):
    """Pass E3 for pixels [lo,hi): returns list of per-pixel byte strings."""
    out = []
    table = entry_tab.copy()
    last_edge = entry_le                      # max edge position < current pixel
    for i in range(lo, hi):
        p = px[i]; q = px[i - 1] if i > 0 else INIT_PREV
        b = bytearray()
        if p == q:
            r = i - last_edge                 # consecutive repeats ending here (last_edge == -1 -> i+1)
            if r % 62 == 0 or i == n - 1:
                b.append(0xC0 | ((r - 1) % 62))
        else:
            pend = (i - 1 - last_edge) % 62   # repeats not yet flushed
            if pend:
                b.append(0xC0 | (pend - 1))
            last_edge = i
            h = int(hash_slots(np.array([p], dtype=np.uint32))[0])
            if table[h] == p:
                b.append(h)
            else:
                table[h] = p
                pr, pg, pb, pa = int(p) & 255, (int(p) >> 8) & 255, (int(p) >> 16) & 255, int(p) >> 24
                qr, qg, qb, qa = int(q) & 255, (int(q) >> 8) & 255, (int(q) >> 16) & 255, int(q) >> 24
                if pa != qa:
                    b += bytes([0xFF, pr, pg, pb, pa])
                else:
                    s8 = lambda v: ((v + 128) & 255) - 128
                    dr, dg, db = s8(pr - qr), s8(pg - qg), s8(pb - qb)
                    drg, dbg = s8(dr - dg), s8(db - dg)
                    if -2 <= dr <= 1 and -2 <= dg <= 1 and -2 <= db <= 1:
                        b.append(0x40 | (dr + 2) << 4 | (dg + 2) << 2 | (db + 2))
                    elif -32 <= dg <= 31 and -8 <= drg <= 7 and -8 <= dbg <= 7:
                        b += bytes([0x80 | (dg + 32), (drg + 8) << 4 | (dbg + 8)])
                    else:
                        b += bytes([0xFE, pr, pg, pb])
        out.append(bytes(b))
    return out


def encode_chunks(px, slab):
    """Chunk bytes (no header/trailer) produced slab by slab with carried state only."""
    px = np.asarray(px, dtype=np.uint32)
    n = len(px)
    tab, valid, le = slab_summaries(px, slab)
    etab, ele = scan_entries(tab, valid, le)
    parts = []
    for s in range(len(le)):
        lo, hi = s * slab, min(n, (s + 1) * slab)
        parts.extend(encode_slab(px, lo, hi, n, etab[s], int(ele[s])))
    return b"".join(parts)
