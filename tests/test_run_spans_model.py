"""dec_segments_rec's pixel sink for images that are not flat (csrc/qoi_decode.hip, round 5), restated lane by lane in Python: a 32-pixel
ring drained in groups of 16, every chunk's first two pixels put unconditionally, and - the part this model is for - QOI_OP_RUNs of twelve
pixels or more taken out of the ring as SPANS that the runs following them directly lengthen (one descriptor per stretch for
dec_expand_runs, at most kSummaryDescs per segment, the run written by the lane itself beyond that).  Whatever the sequence of chunks,
every pixel of the segment must be written exactly once - by the ring or by a span, never both - with the right value, and no span may
cover a pixel the ring still holds.  No GPU; the kernel's own tests are the decode tests of tests/test_gpu_parity.py and the decode fuzz."""
import numpy as np
import pytest

RING, GROUP, LONG_RUN, MAX_DESCS = 32, 16, 12, 32
DRAIN_EVERY = GROUP // 2


class Lane:
    def __init__(self, start, desc_all=True):
        self.fpos = self.ppos = start
        self.ring = [None] * RING
        self.written = {}                 # pixel index -> value, by the lane's own stores
        self.descs = []                   # closed spans (start, length, value)
        self.open = None                  # the span in the making [start, length, value]
        self.desc_all = desc_all

    # ---- LaneWriter ----
    def _store(self, i, v):
        assert i not in self.written, f"pixel {i} written twice"
        self.written[i] = v

    def put(self, px):
        if self.ppos - self.fpos == RING:
            self.drain()
        self.ring[self.ppos % RING] = px
        self.ppos += 1

    def put2n(self, px, n):
        assert self.ppos + 2 - self.fpos <= RING, "put2n needs two free places"
        self.ring[self.ppos % RING] = px
        self.ring[(self.ppos + 1) % RING] = px
        self.ppos += n

    def drain(self):
        while self.fpos % GROUP and self.fpos < self.ppos:
            self._store(self.fpos, self.ring[self.fpos % RING]); self.fpos += 1
        while self.fpos + GROUP <= self.ppos:
            for k in range(GROUP):
                self._store(self.fpos + k, self.ring[(self.fpos + k) % RING])
            self.fpos += GROUP

    def finish(self):
        self.drain()
        while self.fpos < self.ppos:
            self._store(self.fpos, self.ring[self.fpos % RING]); self.fpos += 1

    def splat(self, px, n):
        while self.ppos % 4 and n:
            self.put(px); n -= 1
        self.finish()
        while n >= 4:
            for k in range(4):
                self._store(self.ppos + k, px)
            self.ppos += 4; n -= 4
        self.fpos = self.ppos
        return n

    # ---- one chunk record: `rem` pixels of value px ----
    def step(self, px, rem):
        n2 = min(rem, 2)
        self.put2n(px, n2)
        if rem >= 3:
            rem -= n2
            if rem >= LONG_RUN:
                if self.desc_all:
                    o = self.open
                    if o is not None and self.fpos + n2 == self.ppos and o[0] + o[1] + n2 == self.ppos and o[2] == px:
                        self.ppos += rem; o[1] += rem + n2; self.fpos = self.ppos; rem = 0
                    elif len(self.descs) + 1 < MAX_DESCS:
                        if o is not None:
                            self.descs.append(tuple(o))
                        self.ppos -= n2
                        self.finish()
                        self.open = [self.ppos, rem + n2, px]
                        self.ppos += rem + n2; self.fpos = self.ppos; rem = 0
                    else:
                        rem = self.splat(px, rem)
                else:
                    rem = self.splat(px, rem)
            while rem:
                self.put(px); rem -= 1
            if self.ppos - self.fpos > RING - 2 * DRAIN_EVERY:
                self.drain()

    def run(self, records):
        for i, (px, rem) in enumerate(records):
            if i % DRAIN_EVERY == 0:
                self.drain()                                  # (drain_block: a static number of stores in the kernel)
            self.step(px, rem)
        if self.open is not None:
            self.descs.append(tuple(self.open))
        self.finish()


def _records(rng, kind, n):
    out = []
    v = 1
    for _ in range(n):
        r = rng.random()
        if kind == "sprite" and r < 0.08:                     # a transparent stretch: runs that follow each other directly
            for _ in range(int(rng.integers(1, 30))):
                out.append((v, 62))
            out.append((v, int(rng.integers(1, 63))))
        elif r < (0.5 if kind == "runs" else 0.1):
            out.append((v, int(rng.integers(1, 63))))         # a run of the pixel before
        else:
            v += 1
            out.append((v, 1))                                # a chunk that names a pixel
    return out


@pytest.mark.parametrize("kind", ["photo", "runs", "sprite"])
@pytest.mark.parametrize("desc_all", [True, False])
def test_every_pixel_written_once_by_ring_or_span(kind, desc_all):
    rng = np.random.default_rng(len(kind) * 7 + desc_all)
    spans_seen = merged_seen = overflow_seen = 0
    for trial in range(60):
        start = int(rng.integers(0, 5000))
        recs = _records(rng, kind, int(rng.integers(5, 600)))
        lane = Lane(start, desc_all)
        lane.run(recs)
        # what the chunks say
        want = {}
        pos = start
        for px, rem in recs:
            for _ in range(rem):
                want[pos] = px; pos += 1
        got = dict(lane.written)
        for (s, n, v) in lane.descs:
            assert n >= LONG_RUN + 2, (s, n)                    # (a run takes this path with twelve pixels behind its first two)
            for i in range(s, s + n):
                assert i not in got, f"pixel {i} by a span and by the ring (trial {trial})"
                got[i] = v
        assert got == want, (kind, trial, len(recs))
        assert len(lane.descs) <= MAX_DESCS
        spans_seen += len(lane.descs)
        merged_seen += sum(1 for (_, n, _) in lane.descs if n > 64)
        overflow_seen += len(lane.descs) == MAX_DESCS
    if desc_all and kind != "photo":
        assert spans_seen > 0
    if desc_all and kind == "sprite":
        assert merged_seen > 0                                  # runs that followed each other became one span
    if not desc_all:
        assert spans_seen == 0


def test_more_stretches_than_descriptors():
    """A segment with more long stretches than its summary slot holds descriptors: the rest is written by the lane itself."""
    rng = np.random.default_rng(5)
    recs = []
    v = 0
    for i in range(80):
        v += 1
        recs += [(v, 1), (v, int(rng.integers(20, 63))), (v + 1000, 1)]
    lane = Lane(3, True)
    lane.run(recs)
    assert len(lane.descs) == MAX_DESCS - 1 or len(lane.descs) == MAX_DESCS
    want, pos = {}, 3
    for px, rem in recs:
        for _ in range(rem):
            want[pos] = px; pos += 1
    got = dict(lane.written)
    for (s, n, val) in lane.descs:
        for i in range(s, s + n):
            assert i not in got
            got[i] = val
    assert got == want
