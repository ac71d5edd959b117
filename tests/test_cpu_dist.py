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
