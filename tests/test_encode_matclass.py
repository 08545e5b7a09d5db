"""Host rehearsal of the encoder's matrix-pipe classes (enc_sets CLS 1, 2): the linear forms one v_mfma_i32_32x32x16_i8 hands
every lane, evaluated as the ISA defines the instruction, give the chunk words of qoi.h:438-474."""
import numpy as np

from tools import matclass_model as MM


def _check(px, prev):
    got_w, _, got_h = MM.step_words(px, prev)
    want_w, want_h = MM.reference_words(px, prev)
    assert np.array_equal(got_h, want_h)
    bad = np.nonzero(got_w != want_w)[0]
    assert bad.size == 0, (hex(int(px[bad[0]])), hex(int(prev[bad[0]])), hex(int(got_w[bad[0]])), hex(int(want_w[bad[0]])))


def test_random_pairs():
    rng = np.random.default_rng(7)
    for _ in range(40):
        px = rng.integers(0, 2**32, 64, dtype=np.uint64).astype(np.uint32)
        prev = rng.integers(0, 2**32, 64, dtype=np.uint64).astype(np.uint32)
        _check(px, prev)


def test_small_deltas_every_wrap():
    """deltas around the DIFF / LUMA windows on every channel, at bases that wrap around 0 / 255 and cross 127 / 128 (the
    instruction reads the bytes as signed)"""
    rng = np.random.default_rng(11)
    bases = [0, 1, 2, 3, 30, 31, 33, 120, 126, 127, 128, 129, 130, 200, 250, 253, 254, 255]
    deltas = list(range(-40, 41)) + [-128, -127, 127, 100, -100]
    pairs = []
    for _ in range(6000):
        base = [bases[rng.integers(len(bases))] for _ in range(3)]
        d = [deltas[rng.integers(len(deltas))] for _ in range(3)]
        a = int(rng.integers(0, 256))
        pa = a if rng.random() < 0.9 else int(rng.integers(0, 256))
        prev = base[0] | base[1] << 8 | base[2] << 16 | pa << 24
        px = ((base[0] + d[0]) & 255) | ((base[1] + d[1]) & 255) << 8 | ((base[2] + d[2]) & 255) << 16 | a << 24
        pairs.append((px, prev))
    arr = np.array(pairs, dtype=np.uint32)
    for i in range(0, len(arr) - 63, 64):
        _check(arr[i:i + 64, 0].copy(), arr[i:i + 64, 1].copy())


def test_exhaustive_green_window():
    """every (vg, vr - vg, vb - vg) around the LUMA window: 72 x 20 x 20 cases on a wrapping base"""
    cases = []
    for vg in range(-36, 36):
        for ur in range(-10, 10):
            for ub in range(-10, 10):
                vr, vb = vg + ur, vg + ub
                prev = 250 | 3 << 8 | 128 << 16 | 255 << 24
                px = ((250 + vr) & 255) | ((3 + vg) & 255) << 8 | ((128 + vb) & 255) << 16 | 255 << 24
                cases.append((px, prev))
    while len(cases) % 64:
        cases.append(cases[-1])
    arr = np.array(cases, dtype=np.uint32)
    for i in range(0, len(arr), 64):
        _check(arr[i:i + 64, 0].copy(), arr[i:i + 64, 1].copy())
