"""world_size-2 gloo tests of the multi-process path (CPU): ray sharding, the one-collective
gradient/loss bucket, max-over-ranks timing — what bench.py --gpus N and a training step rely on."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                      WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from nerfactor_amd import dist as nd
    r, w = nd.init_from_env(backend='gloo')
    assert (r, w) == (rank, world) == nd.world()
    # (1) ray sharding: disjoint, contiguous, covers everything (n not divisible by world)
    n = 1001
    rays = torch.arange(n, dtype=torch.float32)[:, None].repeat(1, 3)
    mine = nd.shard(rays)
    lo, hi = nd.shard_range(n)
    gathered = [None] * world
    dist.all_gather_object(gathered, (lo, hi, float(mine.sum())))
    # (2) one flat bucket = all grads + the scalar loss, summed over ranks in ONE call
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(7, 5)), torch.nn.Parameter(torch.zeros(3))]
    grads = [torch.full((7, 5), float(rank + 1)), None]  # a frozen / unused parameter -> zeros
    bucket = nd.FlatBucket(params)
    bucket.pack(grads, scalar=0.25 * (rank + 1))
    views, loss = bucket.all_reduce()
    # (3) wall-clock of a timed region = max over ranks
    tmax = nd.max_over_ranks(1.0 + rank)
    q.put((rank, gathered, views[0].clone().numpy(), views[1].clone().numpy(), float(loss), tmax))
    dist.barrier()
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world = 2
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, gathered, g0, g1, loss, tmax in results:
        spans = sorted((lo, hi) for lo, hi, _ in gathered)
        assert spans[0][0] == 0 and spans[-1][1] == 1001
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) - min(hi - lo for lo, hi in spans) <= 1
        assert abs(sum(s for _, _, s in gathered) - 3 * sum(range(1001))) < 1e-3
        np.testing.assert_allclose(g0, np.full((7, 5), 3.0))   # 1 + 2
        np.testing.assert_allclose(g1, np.zeros(3))
        assert abs(loss - 0.75) < 1e-7                          # 0.25 + 0.5
        assert tmax == 2.0


def test_single_process_defaults():
    from nerfactor_amd import dist as nd
    assert nd.world() == (0, 1)
    assert nd.shard_range(10) == (0, 10)
    assert nd.max_over_ranks(3.5) == 3.5
    b = nd.FlatBucket([torch.nn.Parameter(torch.zeros(4))])
    b.pack([torch.ones(4)], 2.0)
    views, loss = b.all_reduce()
    assert float(loss) == 2.0 and float(views[0].sum()) == 4.0


def _keras_amsgrad_cpu(p, g, m, v, vhat, lr, step, beta1=0.9, beta2=0.999, eps=1e-7):
    """CPU stand-in for nfx_amsgrad_step (tf.keras Adam(amsgrad=True): epsilon outside the sqrt, bias correction in
    the step size) so that optim.AMSGrad's bucket / collective / aliasing logic can run under gloo without a GPU."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    torch.maximum(vhat, v, out=vhat)
    lr_t = lr * (1 - beta2 ** step) ** 0.5 / (1 - beta1 ** step)
    p.sub_(lr_t * m / (vhat.sqrt() + eps))


def _train_worker(rank, world):
    """ranks that build their models from different RNG states must still train ONE model."""
    from nerfactor_amd import dist as nd, ops, optim
    ops.amsgrad_step = _keras_amsgrad_cpu
    torch.manual_seed(100 + rank)                      # deliberately different initial weights per rank
    model = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))
    model.register_buffer('frozen_table', torch.randn(5))
    opt = optim.AMSGrad(model.parameters(), lr=1e-2)
    nd.broadcast_model(model, opt)
    g = torch.Generator().manual_seed(7)
    x_all, y_all = torch.randn(3, 8, 6, generator=g), torch.randn(3, 8, 2, generator=g)
    lo, hi = nd.shard_range(8, rank, world)
    losses = []
    for k in range(3):
        opt.zero_grad()
        per_example = ((model(x_all[k, lo:hi]) - y_all[k, lo:hi]) ** 2).mean(-1)
        weighted = per_example.sum() / 8
        weighted.backward()
        losses.append(opt.step(loss=weighted.detach()))
    # (a) identical replicas
    flat = [torch.empty_like(opt.flat) for _ in range(world)]
    dist.all_gather(flat, opt.flat)
    assert all(torch.equal(flat[0], f) for f in flat[1:])
    tabs = [torch.empty(5) for _ in range(world)]
    dist.all_gather(tabs, model.frozen_table)
    assert torch.equal(tabs[0], tabs[1])
    # (b) the same trajectory as ONE process on the whole batch, started from rank 0's initial weights
    torch.manual_seed(100)
    ref = torch.nn.Sequential(torch.nn.Linear(6, 16), torch.nn.Tanh(), torch.nn.Linear(16, 2))
    ropt = optim.AMSGrad(ref.parameters(), lr=1e-2)
    ref_losses = []
    was = dist.is_initialized
    for k in range(3):
        ropt.zero_grad()
        w = ((ref(x_all[k]) - y_all[k]) ** 2).mean(-1).sum() / 8
        w.backward()
        ropt.bucket.flat[-1] = w.detach()            # single-process step: no collective
        ops.amsgrad_step(ropt.flat, ropt.bucket.flat[:-1], ropt.m, ropt.v, ropt.vhat, ropt.lr, k + 1)
        ref_losses.append(float(w))
    assert torch.allclose(opt.flat, ropt.flat, rtol=0, atol=2e-6), (opt.flat - ropt.flat).abs().max()
    # (c) ADVICE r01 (medium): the returned losses are values, not aliases of the bucket's last slot
    got = [float(l) for l in losses]
    assert np.allclose(got, ref_losses, rtol=1e-5) and len(set(got)) == 3, (got, ref_losses)


def test_two_ranks_train_one_model():
    from tests import mp_util
    mp_util.run_workers(_train_worker, world=2)


class _MockRenderModel(torch.nn.Module):
    """CPU stand-in for a render plugin: per-ray outputs that depend only on that ray's inputs (what the libnfx kernels
    guarantee), returned through the plugin contract; vis_rows / write_vis are the product's own (models/base.py)."""
    def __init__(self):
        super().__init__()
        from nerfactor_amd.nerfactor.models.base import Model
        self.vis_rows, self.write_vis = Model.vis_rows, Model.write_vis

    def forward(self, batch, mode='test', relight_probes=False):
        id_, hw, rayo, rayd, rgb = batch
        g = torch.Generator().manual_seed(1)
        w = torch.randn(3, 3, generator=g)
        pred_rgb = torch.sigmoid(rayd @ w + rayo.sum(1, keepdim=True))
        to_vis = {'id': id_, 'hw': hw, 'gt_rgb': rgb, 'pred_rgb': pred_rgb, 'pred_normal': torch.tanh(rayd),
                  'pred_lvis': torch.sigmoid(rayd @ torch.randn(3, 16, generator=g)),
                  'pred_albedo': pred_rgb * 0.5}
        if relight_probes:
            to_vis['pred_rgb_probes'] = torch.stack((pred_rgb, 1 - pred_rgb), 1)
        return {}, rgb, {}, to_vis


def _view_batch():
    h, w = 7, 9     # 63 rays: not divisible by 2
    g = torch.Generator().manual_seed(5)
    n = h * w
    return (['test_000'] * n, torch.tensor([[h, w]] * n, dtype=torch.int32), torch.randn(n, 3, generator=g),
            torch.randn(n, 3, generator=g), torch.rand(n, 3, generator=g))


def _render_worker(rank, world):
    from nerfactor_amd.nerfactor.util import shard
    shard.render_view(_MockRenderModel(), _view_batch(), os.environ['NFX_TEST_OUTDIR'], mode='test', relight_probes=True)


def test_rays_within_view_sharding_stitches_the_one_rank_image(tmp_path):
    """VERDICT r01 #7 / SURVEY §8e: the render drivers shard the rays OF EACH VIEW over the ranks; rank 0 receives
    uint8 rows only and must write exactly the files a single process writes."""
    from nerfactor_amd.nerfactor.util import shard
    from tests import mp_util
    one, two = str(tmp_path / 'one'), str(tmp_path / 'two')
    shard.render_view(_MockRenderModel(), _view_batch(), one, mode='test', relight_probes=True)
    os.environ['NFX_TEST_OUTDIR'] = two
    try:
        mp_util.run_workers(_render_worker, world=2)
    finally:
        del os.environ['NFX_TEST_OUTDIR']
    files = sorted(os.path.relpath(os.path.join(d, f), one) for d, _, fs in os.walk(one) for f in fs)
    assert 'pred_rgb.png' in files and os.path.join('pred_rgb_probes', '0001.png') in files and 'metadata.json' in files
    assert files == sorted(os.path.relpath(os.path.join(d, f), two) for d, _, fs in os.walk(two) for f in fs)
    for f in files:
        assert open(os.path.join(one, f), 'rb').read() == open(os.path.join(two, f), 'rb').read(), f


def test_graphed_step_only_captures_static_single_process_batches():
    """optim.GraphedTrainStep captures a step only when nothing in it depends on data-dependent shapes: batches whose
    alpha the dataset tagged foreground-only (or NeRF batches, which have no compaction), single process."""
    import types
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
    opt = types.SimpleNamespace(flat=torch.zeros(4))
    g = optim.GraphedTrainStep(model=None, optimizer=opt, global_bs=8)
    n = 8
    surf = lambda alpha: (None, None, torch.zeros(n, 3), torch.zeros(n, 3), torch.zeros(n, 3), alpha, torch.zeros(n, 3),
                          torch.zeros(n, 3), torch.zeros(n, 16))
    assert not g._capturable(surf(torch.ones(n, 1)))                        # untagged: may hold background rays
    assert g._capturable(surf(mark_all_foreground(torch.ones(n, 1))))
    assert g._capturable((None, None, torch.zeros(n, 3), torch.zeros(n, 3), torch.zeros(n, 3)))   # NeRF batch
    a, b = surf(mark_all_foreground(torch.ones(n, 1))), surf(mark_all_foreground(torch.ones(n, 1)))
    assert g._key(a) == g._key(b) and g._key(a) != g._key(surf(mark_all_foreground(torch.ones(4, 1))))


def test_train_step_hands_back_detached_visualisation_tensors():
    """optim._detached: what a training step returns for visualisation must not keep the step's autograd graph (and
    with it the parameters' AccumulateGrad nodes) alive across steps — profiles/HISTORY.md §3b, the hipGraph capture."""
    from nerfactor_amd import optim
    w = torch.ones(3, requires_grad=True)
    to_vis = {'id': ['a'], 'hw': torch.tensor([[2, 2]]), 'pred_rgb': w * 2., 'gt_rgb': torch.zeros(3)}
    out = optim._detached(to_vis)
    assert out['id'] == ['a'] and out['pred_rgb'].grad_fn is None and not out['pred_rgb'].requires_grad
    assert torch.equal(out['pred_rgb'], to_vis['pred_rgb']) and out['gt_rgb'] is not None
    assert optim._detached(None) is None


def _gather_cat_worker(rank, world):
    from nerfactor_amd import dist as nd
    from nerfactor_amd.nerfactor.util import shard as shardutil
    n = 1001                                    # not divisible by 3: ragged shards
    full = (torch.arange(n * 4 * 3) % 251).to(torch.uint8).reshape(n, 4, 3)
    lo, hi = nd.shard_range(n)
    rows = {'hw': (7, 143), 'id': 'v', 'pred_rgb': full[lo:hi].clone(), 'flat': full[lo:hi, 0, 0].clone()}
    out = shardutil.gather_rows(rows)
    if rank == 0:                               # rank 0 alone holds the stitched view, in ray order
        assert torch.equal(out['pred_rgb'], full) and torch.equal(out['flat'], full[:, 0, 0])
        assert out['hw'] == (7, 143) and out['id'] == 'v'
    else:
        assert out is None
    got = nd.gather_cat(full[lo:hi].clone(), dst=1)
    assert torch.equal(got, full if rank == 1 else full[lo:hi])


def test_rank0_gather_of_uint8_rows_is_point_to_point_and_ragged():
    """dist.gather_cat / util.shard.gather_rows ("rank 0 optionally gathers uint8 frames only"): three
    ranks with ragged shards; only the destination rank assembles the view."""
    from tests.mp_util import run_workers
    run_workers(_gather_cat_worker, world=3)


def _forced_group_worker(rank, world, q):
    from nerfactor_amd import dist as nfx_dist
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT'):
        os.environ.pop(k, None)
    nfx_dist.init_from_env(backend='gloo', force=True)
    ok = dist.is_initialized() and dist.get_world_size() == 1 and dist.get_backend() == 'gloo'
    # a gloo group of one rank keeps the shortcut: only an nccl (RCCL) group of one rank runs the collectives for real
    shortcut = not nfx_dist.run_collectives_on_one_rank()
    params = [torch.arange(6, dtype=torch.float32).reshape(2, 3), torch.ones(4)]
    bucket = nfx_dist.FlatBucket(params)
    bucket.pack([p * 2 for p in params], 0.5)
    before = bucket.flat.clone()
    views, total = bucket.all_reduce()
    q.put((ok, shortcut, bool(torch.equal(bucket.flat, before)), float(total), nfx_dist.max_over_ranks(3.5)))
    dist.destroy_process_group()


def test_forced_one_rank_group_initialises_and_keeps_values():
    """dist.init_from_env(force=True) (bench.py --force-group, the RCCL tests): a single process gets a process group of one
    rank on a free port; on gloo the step's collectives keep their one-rank shortcut, the bucket is unchanged either way."""
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    p = ctx.Process(target=_forced_group_worker, args=(0, 1, q))
    p.start()
    ok, shortcut, same, total, mx = q.get(timeout=120)
    p.join(60)
    assert ok and shortcut and same and total == 0.5 and mx == 3.5
