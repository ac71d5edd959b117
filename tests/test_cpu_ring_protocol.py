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
