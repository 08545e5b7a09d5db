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


# ------------------------------------------------------------------------------------------------------------------------
# State look-back over the sets of an image (csrc/qoi_encode.hip: g2_entry_state, enc_sets<ENTRY 2 / 3>): the protocol, stated
# as a model that a scheduler can interleave any way it likes.  A set owns 65 granules - one per table slot, one for the last
# edge - each written whole (8 bytes on the device) and each in one of three states: EMPTY (not of this call), LOCAL (what the
# set's own pixels say: the last edge pixel it holds for the slot, if any), INCLUSIVE (what the table holds BEHIND the set,
# whatever came before it).  Nothing orders the 65 stores of a publication against each other or against a reader.
# ------------------------------------------------------------------------------------------------------------------------
EMPTY, LOCAL, INCL = 0, 1, 2


def set_summary(px, lo, hi):
    """Last edge pixel per slot (+ valid) and last edge position of the pixels [lo, hi); the pixel in front of lo decides whether lo is an edge."""
    tab = np.zeros(64, dtype=np.uint32); valid = np.zeros(64, dtype=bool); le = -1
    if hi <= lo:
        return tab, valid, le
    seg = px[lo:hi]
    prev = np.concatenate([[px[lo - 1] if lo > 0 else INIT_PREV], seg[:-1]])
    idx = np.nonzero(seg != prev)[0]
    if len(idx):
        le = int(idx[-1]) + lo
        tab[hash_slots(seg[idx])] = seg[idx]
        valid[hash_slots(seg[idx])] = True
    return tab, valid, le


def tail_first_summary(px, lo, hi, tail):
    """What a set publishes first: its last `tail` pixels alone where they write all 64 slots and hold an edge, else those merged with the
    pixels in front of them (the tail's words win).  Returns (tab, valid, le, front_walked)."""
    cut = max(lo, hi - tail)
    t_tab, t_valid, t_le = set_summary(px, cut, hi)
    if cut == lo or (t_valid.all() and t_le >= 0):
        return t_tab, t_valid, t_le, not (t_valid.all() and t_le >= 0)
    f_tab, f_valid, f_le = set_summary(px, lo, cut)
    return np.where(t_valid, t_tab, f_tab), t_valid | f_valid, (t_le if t_le >= 0 else f_le), True


def simulate_state_lookback(px, set_px, rng, tail=None, fast=(), window=8, max_steps=2_000_000, stats=None):
    """Runs every set of the image as a coroutine of single granule reads / writes under a random scheduler (sets START in order, as the
    tickets hand them out; after that any interleaving) and returns, per set that looked back, the entry table and entry edge it resolved.
    `fast`: sets that know their entry state by other means (ENTRY 3: the look-back window) - they publish INCLUSIVE granules only, at the
    very end, with every slot marked valid."""
    px = np.asarray(px, dtype=np.uint32)
    n = len(px)
    ns = (n + set_px - 1) // set_px
    state = np.zeros((ns, 65), dtype=np.int8)           # [set][slot | 64 = last edge]
    word = np.zeros((ns, 65), dtype=np.int64)
    valid = np.zeros((ns, 65), dtype=bool)
    resolved = {}
    truth_tab, truth_valid, truth_le = slab_summaries(px, set_px)
    truth_etab, truth_ele = scan_entries(truth_tab, truth_valid, truth_le)

    def put(k, s, st, w, v):
        assert st > state[k, s], "a granule only moves forward"
        state[k, s] = st; word[k, s] = w; valid[k, s] = v

    def run(k):
        lo, hi = k * set_px, min(n, (k + 1) * set_px)
        if k in fast:
            # (the encode runs here; the table behind the set = entry merged with what the set wrote)
            for _ in range(int(rng.integers(0, 40))):
                yield
            after = np.where(truth_valid[k], truth_tab[k], truth_etab[k])
            le_after = max(int(truth_le[k]), int(truth_ele[k]))
            for s in rng.permutation(65):
                put(k, s, INCL, (le_after + 1) if s == 64 else int(after[s]), True)
                yield
            return
        tab, val, le, _ = tail_first_summary(px, lo, hi, tail if tail else hi - lo)
        first = INCL if (val.all() and le >= 0) else LOCAL
        if stats is not None:
            stats["inclusive_at_once" if first == INCL else "local_first"] = stats.get("inclusive_at_once" if first == INCL else "local_first", 0) + 1
        for s in rng.permutation(65):
            put(k, s, first, (le + 1) if s == 64 else int(tab[s]), (le >= 0) if s == 64 else bool(val[s]))
            yield
        ent_w = np.zeros(65, dtype=np.int64); ent_v = np.zeros(65, dtype=bool); done = np.zeros(65, dtype=bool)
        j0 = k - 1
        while j0 >= 0 and not done.all():
            # one poll: `window` sets, all 65 granules of each must be of this call before any is used
            js = [j for j in range(j0, j0 - window, -1)]
            while True:
                snap = [(state[j].copy(), word[j].copy(), valid[j].copy()) if j >= 0 else None for j in js]
                yield
                if all(sn is None or (sn[0] != EMPTY).all() for sn in snap):
                    break
                if stats is not None:
                    stats["polls_again"] = stats.get("polls_again", 0) + 1
            if stats is not None and any(sn is not None and len(set(sn[0].tolist())) > 1 for sn in snap):
                stats["sets_seen_half_and_half"] = stats.get("sets_seen_half_and_half", 0) + 1
            for sn in snap:
                for s in range(65):
                    if done[s]:
                        continue
                    if sn is None:                        # in front of set 0: the zeroed table, no edge - inclusive
                        done[s] = True
                    elif sn[0][s] == INCL or sn[2][s]:
                        ent_w[s] = sn[1][s]; ent_v[s] = sn[2][s]; done[s] = True
            j0 -= window
        e_tab = np.where(ent_v[:64], ent_w[:64], 0).astype(np.uint32)
        e_le = int(ent_w[64]) - 1 if ent_v[64] else -1
        resolved[k] = (e_tab, e_le)
        for s in rng.permutation(65):
            if s == 64:
                inc = le if le >= 0 else e_le
                put(k, s, INCL, inc + 1, inc >= 0) if state[k, s] != INCL else None
            else:
                if state[k, s] != INCL:
                    put(k, s, INCL, int(tab[s]) if val[s] else int(e_tab[s]), bool(val[s] or ent_v[s]))
            yield

    live, started, steps = [], 0, 0
    while started < ns or live:
        if started < ns and (not live or rng.random() < 0.15):
            live.append(run(started)); started += 1
        g = live[int(rng.integers(0, len(live)))]
        try:
            next(g)
        except StopIteration:
            live.remove(g)
        steps += 1
        assert steps < max_steps, "no progress: the protocol would hang"
    return resolved, truth_etab, truth_ele
