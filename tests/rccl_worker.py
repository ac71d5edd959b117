"""One process, one GPU: the training step's collectives on RCCL with a process group of ONE rank (SURVEY.md §4 item 4,
VERDICT r05 "missing" #2).  Run by tests/test_gpu_rccl.py as a subprocess (a process group is process-wide state, and an
RCCL problem must not take the whole pytest session down):

    python tests/rccl_worker.py bucket | step | graph

prints one JSON object.  Every case first runs WITHOUT a group, then initialises `backend='nccl', world_size=1` and runs
again; a sum over one rank must leave every bit where it was."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def batches(cuda, n=256, k=4):
    from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
    rng = np.random.default_rng(7)
    out = []
    for i in range(k):
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)
        xyz = t(rng.uniform(-1, 1, size=(n, 3)))
        nrm = torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1)
        cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
        out.append((['view%d' % i] * n, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))),
                    mark_all_foreground(torch.ones(n, 1, device=cuda)), xyz, nrm, t(rng.uniform(size=(n, 512)))))
    return out


def train(cuda, graph, steps=10, name='nerfactor_microfacet', n=256, capture_collective=None):
    from nerfactor_amd import dist as nfx_dist, optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    torch.manual_seed(11)
    cfg = make_config(name, xyz_jitter_std='0.01', shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='')
    model = get_model_class(name)(cfg).to(cuda)
    opt = optim.make_optimizer(model, cfg)
    nfx_dist.broadcast_model(model, opt)
    step = optim.GraphedTrainStep(model, opt, n, warmup=2, capture_collective=capture_collective) if graph else \
        (lambda b: optim.train_step(model, b, opt, n))
    bs = batches(cuda, n)
    losses = [step(bs[i % len(bs)])[0] for i in range(steps)]
    model.flush_numerics(block=True)
    torch.cuda.synchronize()
    graphs = len(step.graphs) if graph else 0
    return torch.stack(losses).cpu(), opt.flat.clone().cpu(), opt.vhat.clone().cpu(), graphs


def init_group(cuda):
    from nerfactor_amd import dist as nfx_dist
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        os.environ.pop(k, None)
    nfx_dist.init_from_env(backend='nccl', device=cuda, force=True)
    assert dist.is_initialized() and dist.get_world_size() == 1 and dist.get_backend() == 'nccl'
    assert nfx_dist.run_collectives_on_one_rank()


def count_all_reduces():
    """Wraps torch.distributed.all_reduce: how often the step really issued the collective."""
    calls = {'n': 0}
    orig = dist.all_reduce

    def counted(*a, **kw):
        calls['n'] += 1
        return orig(*a, **kw)
    dist.all_reduce = counted
    return calls


def main(case):
    from nerfactor_amd import build
    build.build()
    from nerfactor_amd import dist as nfx_dist
    cuda = torch.device('cuda', 0)
    torch.cuda.set_device(cuda)
    out = {"case": case}
    if case == 'bucket':
        torch.manual_seed(3)
        params = [torch.randn(5, 7, device=cuda), torch.randn(130, device=cuda), torch.randn(1, device=cuda)]
        grads = [torch.randn_like(p) for p in params]
        bucket = nfx_dist.FlatBucket(params)
        bucket.pack(grads, 0.625)
        before = bucket.flat.clone()
        calls = count_all_reduces()
        bucket.all_reduce()
        out["collectives_without_group"] = calls['n']
        init_group(cuda)
        views, total = bucket.all_reduce()                      # on the current stream
        torch.cuda.synchronize()
        out["collectives_current_stream"] = calls['n']
        out["unchanged_current_stream"] = bool(torch.equal(bucket.flat, before)) and float(total) == 0.625
        # side stream: work queued on the current stream BEFORE the call must be seen by the collective, work queued AFTER
        # it must see the collective's result (FlatBucket.all_reduce's stream semantics)
        side = torch.cuda.Stream(device=cuda)
        big = torch.randn(1 << 24, device=cuda)
        for _ in range(8):                                       # keep the current stream busy while the bucket is re-filled
            big = big * 1.0001
        bucket.pack([g * 2 for g in grads], 1.25)
        want = bucket.flat.clone()
        views, total = bucket.all_reduce(stream=side)
        after = bucket.flat * 1.0                                # queued after the call on the current stream
        torch.cuda.synchronize()
        out["collectives_side_stream"] = calls['n']
        out["unchanged_side_stream"] = bool(torch.equal(after, want)) and bool(torch.equal(views[1], grads[1] * 2))
        out["max_over_ranks"] = nfx_dist.max_over_ranks(1.5, device=cuda)
        out["sum_over_ranks"] = nfx_dist.sum_over_ranks(torch.tensor(2.5, device=cuda))
    elif case in ('step', 'graph'):
        graph = case == 'graph'
        l0, p0, v0, g0 = train(cuda, graph)
        init_group(cuda)
        calls = count_all_reduces()
        l1, p1, v1, g1 = train(cuda, graph, capture_collective=True)
        out.update(collectives=calls['n'], graphs=[g0, g1], finite=bool(torch.isfinite(l1).all()),
                   losses_equal=bool(torch.equal(l0, l1)), params_equal=bool(torch.equal(p0, p1)),
                   vhat_equal=bool(torch.equal(v0, v1)), loss_first=float(l1[0]), loss_last=float(l1[-1]))
    else:
        raise SystemExit("unknown case " + case)
    out["backend"] = dist.get_backend()
    dist.destroy_process_group()
    print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main(sys.argv[1])
