"""Rays-within-view sharding for the render drivers (the reference's analogue is MirroredStrategy
splitting the flat per-ray batch over replicas, trainvali.py:85,100): rank r renders the contiguous ray range
[r n / N, (r + 1) n / N) of EVERY view — so 4 test views keep 8 GPUs busy — quantises its own rows, and only uint8 rows
travel, point to point, to rank 0 (dist.gather_cat: send / recv into rank 0's buffer), which writes the images.  No
collective on the data path."""
import torch

from ... import dist as nfx_dist


def shard_batch(batch):
    """This rank's contiguous ray shard of every per-ray field of a flat batch tuple (lists of ids included)."""
    rank, ws = nfx_dist.world()
    if ws == 1:
        return batch
    n = len(batch[0]) if batch[0] is not None else next(len(x) for x in batch if x is not None)
    lo, hi = nfx_dist.shard_range(n, rank, ws)

    def part(x):
        y = x[lo:hi]
        if getattr(x, '_nfx_all_foreground', False):    # a slice of foreground-only rays is foreground-only
            y._nfx_all_foreground = True
        return y
    return tuple(None if x is None else part(x) for x in batch)


def shard_rows(t, n_total):
    """The rows of a FULL-view per-ray tensor (e.g. an albedo override map) that belong to this rank."""
    rank, ws = nfx_dist.world()
    if ws == 1 or t is None or not isinstance(t, torch.Tensor) or t.dim() == 0 or t.shape[0] != n_total:
        return t
    lo, hi = nfx_dist.shard_range(n_total, rank, ws)
    return t[lo:hi]


def gather_rows(rows):
    """Rank 0: the per-ray uint8 rows of every rank concatenated in rank order (= ray order); other ranks: None."""
    rank, ws = nfx_dist.world()
    if ws == 1:
        return rows
    out = {}
    for k in sorted(rows):
        v = rows[k]
        out[k] = nfx_dist.gather_cat(v) if isinstance(v, torch.Tensor) else v
    return out if rank == 0 else None


def render_view(model, batch, outdir, mode='test', **call_kwargs):
    """model(batch shard, mode, **call_kwargs) on this rank's rays of one view -> PNGs written by rank 0."""
    _, _, _, to_vis = model(shard_batch(batch), mode=mode, **call_kwargs)
    rows = gather_rows(model.vis_rows(to_vis))
    if rows is not None:
        model.write_vis(rows, outdir)
    return to_vis
