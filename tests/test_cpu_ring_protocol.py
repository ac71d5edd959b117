"""The weight-ring protocols of the NeRF MLP kernels (nerf_mlp_v6.hip) as a happens-before model.

Four waves walk the 78 chunks of a network in lock-step tiles; the only cross-wave ordering is the workgroup barrier
at the end of every tile.  An event of wave a in tile i is ordered before an event of wave b in tile j iff i < j
(barrier i lies between them) — within one tile nothing is ordered across waves.  For every chunk the model checks
  RAW  every wave's part of the chunk has landed in LDS, and a barrier has passed, before ANY wave reads it,
  WAR  every wave has finished reading a slot's previous occupant, and a barrier has passed, before ANY wave may
       start overwriting the slot,
for the shipped protocols and shows that the checks do catch the race variant 7 had while it was being written
(3-slot ring with a fetch distance of 3)."""
import pytest

N_CHUNKS = 78


def violations(ring, dist, landed_after, first_read_before_barrier_of=-1):
    """ring: slots; chunk c lives in slot c % ring.  dist: chunk k + dist is issued (DMA) or stored (register-staged)
    by each wave during tile k.  landed_after: a wave's own part of the chunk issued in tile k is guaranteed in LDS
    before the barrier of tile k + landed_after.  Reads of chunk c: the whole of tile c, plus the pre-read of its
    first fragments at the end of tile c - 1 (before that tile's barrier)."""
    out = []
    for c in range(3 * N_CHUNKS):                  # three passes: the wrap-around is part of the protocol
        issue_tile = c - dist
        landed_tile = issue_tile + landed_after    # guaranteed before the barrier that ends this tile
        first_read_tile = c + first_read_before_barrier_of   # the pre-read, before barrier c - 1
        if not landed_tile < first_read_tile:
            out.append(('RAW', c))
        prev = c - ring                            # previous occupant of the slot, last read during tile prev
        if not prev < issue_tile:
            out.append(('WAR', c))
    return out


@pytest.mark.parametrize('name,kw', [
    ('variant 6: 3 slots, chunk k+2 stored at the end of tile k', dict(ring=3, dist=2, landed_after=0)),
    ('variant 7: 6 slots, chunk k+3 issued in tile k, vmcnt leaves one chunk in flight', dict(ring=6, dist=3, landed_after=1)),
    ('variant 7, NFX_V7_DIST=4: two chunks in flight', dict(ring=6, dist=4, landed_after=2)),
    ('variant 8: 3 slots, fetched in tile k-1, stored at the end of tile k', dict(ring=3, dist=2, landed_after=0)),
])
def test_shipped_ring_protocols_are_race_free(name, kw):
    assert violations(**kw) == [], name


def test_the_model_catches_the_known_bad_configurations():
    # the first version of variant 7: 3-slot ring, distance 3 -> the DMA overwrites the slot tile k is reading
    assert ('WAR', 3) in violations(ring=3, dist=3, landed_after=1)
    # waiting only at the end of the tile that precedes the first use: other waves' parts may still be in flight
    assert any(v[0] == 'RAW' for v in violations(ring=6, dist=3, landed_after=2))
    # a ring as small as the fetch distance can never work
    assert violations(ring=4, dist=4, landed_after=2)


# ----------------------------------------------------------------------------------------------------------------------
# The NeRF backward's weight ring (nerf_bwd.hip, namespace nring): hand-counted `s_waitcnt vmcnt` on gfx9, where loads
# AND stores retire through one in-order counter.  The model replays one wave's VMEM instruction stream for one row tile
# and checks that every wait (a) forces the chunk it is there for to have landed, (b) leaves everything in flight that
# it may — it is exact, not merely safe — and (c) that the 6-bit counter is never asked to hold more than 63.
# ----------------------------------------------------------------------------------------------------------------------
import os
import re

N_SEQ, EPI_STORES = 152, 8


def nring_used(i):
    """fragments chunk i of the tile's sequence needs (nerf_bwd.hip:Cfg::pieces; nerf_train_layout.hpp)"""
    j = i - 76
    return (4 if i < 8 else 16 if i < 40 else 20 if i < 48 else 16 if i < 72 else 18 if i < 76
            else 1 if j < 4 else 8 if j < 12 else 17 if j < 20 else 16)


def nring_pieces(i, nw):
    return 0 if i < 0 or i >= N_SEQ else -(-nring_used(i) // nw)


def nring_allow(i, nw, d):
    n = EPI_STORES * (0 if i < 0 else min(i, d - 1))
    return n + sum(nring_pieces(j, nw) for j in range(i + 2, i + d + 1))


def replay_nring(nw, d, allow=nring_allow):
    """-> (problems, max counter value).  The stream of one wave: priming fetches, then per chunk
    [fetch chunk i + d] [MFMAs] [wait] [barrier] [8 epilogue stores]."""
    stream, problems, peak = [], [], 0          # stream: ('dma', chunk) | ('st', chunk), oldest first; retired ones removed

    def wait(k, must_have_landed, where):
        nonlocal stream
        if len(stream) > k:
            stream = stream[len(stream) - k:]    # in-order counter: only the k youngest may still be in flight
        if any(op == ('dma', must_have_landed) for op in stream):
            problems.append(('chunk %d may not have landed' % must_have_landed, where))

    def issue(op, n):
        nonlocal peak
        stream.extend([op] * n)
        peak = max(peak, len(stream))

    for f in range(d):
        issue(('dma', f), nring_pieces(f, nw))
    wait(allow(-1, nw, d), 0, 'priming')
    for i in range(N_SEQ):
        issue(('dma', i + d), nring_pieces(i + d, nw))
        before = list(stream)
        k = allow(i, nw, d)
        if i + 1 < N_SEQ:
            wait(k, i + 1, 'chunk %d' % i)
            # exactness: one more retired instruction than necessary would be a stall the schedule does not need
            need = [op for op in before if op == ('dma', i + 1)]
            if need and len(before) > k and before[len(before) - k - 1] != ('dma', i + 1):
                problems.append(('wait behind chunk %d retires more than chunk %d' % (i, i + 1), k))
        issue(('st', i), EPI_STORES)
    return problems, peak


@pytest.mark.parametrize('nw', [4, 8])
@pytest.mark.parametrize('d', [2, 3, 4, 5])
def test_nerf_backward_ring_waits_are_exact_and_fit_the_counter(nw, d):
    problems, peak = replay_nring(nw, d)
    assert problems == []
    # the most the counter would hold if nothing retired before a wait forces it to.  The shipped form (8 waves) stays
    # below the 6-bit limit everywhere; with 4 waves and a distance of 5 the enc[5] / rgb_out[0] / D3 regions (5 pieces
    # per chunk) reach 52 allowed + 8 stores + 5 pieces = 65: there the hardware holds the wave's next VMEM issue until
    # two older instructions have retired — a possible stall, never a miscount (the waits stay exact)
    assert peak <= 63 or (nw, d, peak) == (4, 5, 65), peak
    # slot reuse: chunk F goes to slot F % (d + 1) while chunk F - d is consumed; the slot's previous occupant F - d - 1
    # was last read one chunk earlier, and that chunk's closing barrier lies in between
    for f in range(d + 1, N_SEQ):
        assert f - (d + 1) < f - d


def test_nerf_backward_ring_model_catches_a_miscounted_wait():
    # one store too many allowed in flight: the chunk the wait is there for may still be on its way
    loose = lambda i, nw, d: nring_allow(i, nw, d) + 1
    assert replay_nring(4, 5, allow=loose)[0]
    # one too few: safe, but no longer exact
    tight = lambda i, nw, d: max(nring_allow(i, nw, d) - 1, 0)
    assert any('retires more' in p[0] for p in replay_nring(8, 5, allow=tight)[0])
    # a fetch distance of 6 does not fit the 6-bit counter with 4 waves (the static_assert of Cfg<4>)
    assert replay_nring(4, 6)[1] > 63


def test_nerf_backward_ring_model_is_the_kernel_source():
    """The constants restated above are the ones in nerf_bwd.hip (a changed kernel must change this model)."""
    src = open(os.path.join(os.path.dirname(__file__), '..', 'nerfactor_amd', 'csrc', 'nerf_bwd.hip')).read()
    assert re.search(r'kSeq = 152, kD = NFX_NRING_D, kR = kD \+ 1, kEpiStores = 8;', src)
    assert '#define NFX_NRING_D 5' in src
    assert ('const int used = i < 8 ? 4 : i < 40 ? 16 : i < 48 ? 20 : i < 72 ? 16 : i < 76 ? 18' in src
            and ': j < 4 ? 1 : j < 12 ? 8 : j < 20 ? 17 : 16;' in src)
    assert 'int n = kEpiStores * (i < 0 ? 0 : i < kD - 1 ? i : kD - 1);' in src
    assert 'for (int j = i + 2; j <= i + kD; ++j) n += pieces(j);' in src
    # eight dword stores per epilogue: feat_store.hpp:store_tile issues two st32 per even j < 8
    fs = open(os.path.join(os.path.dirname(__file__), '..', 'nerfactor_amd', 'csrc', 'feat_store.hpp')).read()
    body = fs[fs.index('void store_tile('):]
    body = body[:body.index('\n}\n')]
    assert 'for (int j = 0; j < 8; j += 2)' in body and body.count('st32(') == 2
