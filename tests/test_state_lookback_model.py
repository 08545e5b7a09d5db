"""The encoder's state look-back over the sets of a flagged image (csrc/qoi_encode.hip: g2_entry_state, enc_sets<ENTRY 2 / 3>) as a model
(tools/encode_model.py): every set a coroutine of single 8-byte granule reads and writes, a random scheduler in between.  Whatever the
interleaving - a reader may find a set in front of it half LOCAL and half INCLUSIVE, or not there at all - each set must resolve exactly the
table and the last edge the sequential encoder holds at its first pixel (qoi.h:393-399, 425-436), and the scheduler must never run out of
runnable work.  No GPU; the kernels' own tests are tests/test_gpu_parity.py (flat frames, granules across calls, the encoder fuzz)."""
import numpy as np
import pytest

from tools import encode_model as M


def _image(rng, n, kind):
    """Pixels with long flat stretches (few edges: sets whose summaries leave slots open) between busy ones (sets that write all 64 slots)."""
    pal = rng.integers(0, 2 ** 32, size=6, dtype=np.uint64).astype(np.uint32)
    px = np.empty(n, dtype=np.uint32)
    i = 0
    while i < n:
        m = min(n - i, int(rng.integers(20, 700)))
        if kind == "flat" or (kind == "mixed" and rng.random() < 0.5):
            px[i:i + m] = pal[int(rng.integers(0, 6))]
        else:
            px[i:i + m] = rng.integers(0, 2 ** 32, size=m, dtype=np.uint64).astype(np.uint32)
        i += m
    return px


def test_tail_first_summary_is_the_whole_summary():
    """A set's last pixels alone where they write every slot, merged with the pixels in front of them where they do not: the same table,
    the same valid slots, the same last edge as one walk over the whole set."""
    rng = np.random.default_rng(11)
    walked = {True: 0, False: 0}
    for trial in range(300):
        n = int(rng.integers(50, 4000))
        px = _image(rng, n, ("flat", "busy", "mixed")[trial % 3])
        lo = int(rng.integers(0, n - 1)); hi = int(rng.integers(lo + 1, n + 1))
        tail = int(rng.integers(1, 1500))
        tab, valid, le, front = M.tail_first_summary(px, lo, hi, tail)
        w_tab, w_valid, w_le = M.set_summary(px, lo, hi)
        assert (valid == w_valid).all() and le == w_le and (tab[valid] == w_tab[w_valid]).all(), (trial, lo, hi, tail)
        walked[bool(front)] += 1
    assert walked[True] > 30 and walked[False] > 30, walked          # both branches seen


@pytest.mark.parametrize("kind", ["flat", "mixed", "busy"])
@pytest.mark.parametrize("window", [1, 4, 8])
def test_every_set_resolves_the_sequential_entry_state(kind, window):
    rng = np.random.default_rng(1000 + window + len(kind))
    looked = 0
    stats = {}
    for trial in range(12):
        n = int(rng.integers(600, 5000))
        set_px = int(rng.choice([64, 256, 700]))
        px = _image(rng, n, kind)
        if trial % 4 == 0:
            px[: int(rng.integers(1, n // 2))] = M.INIT_PREV            # the image opens with the start value: no edge for a long while
        resolved, etab, ele = M.simulate_state_lookback(px, set_px, rng, tail=int(rng.choice([set_px, set_px // 2, 16])), window=window, stats=stats)
        assert len(resolved) == len(ele)
        for k, (tab, le) in resolved.items():
            assert (tab == etab[k]).all() and le == ele[k], (kind, window, trial, k)
        looked += len(resolved)
    assert looked > 50
    # the schedules were not kind: readers polled again for sets that were not there yet and met sets half LOCAL, half INCLUSIVE
    assert stats.get("polls_again", 0) > 0 and stats.get("local_first", 0) > 0, stats
    if kind != "flat":
        assert stats.get("inclusive_at_once", 0) > 0 and stats.get("sets_seen_half_and_half", 0) > 0, stats


def test_sets_that_publish_inclusive_only_at_their_end():
    """ENTRY 3: sets whose look-back window gave them their entry state publish nothing but INCLUSIVE granules, late (behind their
    encoding), every slot marked valid; the sets that look back over them wait for that and still resolve the sequential state."""
    rng = np.random.default_rng(77)
    for trial in range(10):
        n = int(rng.integers(1500, 6000)); set_px = 512
        px = _image(rng, n, "mixed")
        ns = (n + set_px - 1) // set_px
        fast = {k for k in range(ns) if rng.random() < 0.5}
        resolved, etab, ele = M.simulate_state_lookback(px, set_px, rng, tail=400, fast=fast, window=8)
        assert set(resolved) == set(range(ns)) - fast
        for k, (tab, le) in resolved.items():
            assert (tab == etab[k]).all() and le == ele[k], (trial, k)
