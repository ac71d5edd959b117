#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[3], SURVEY.md §8d "C4"): training steps — 1024 rays per GPU per
step (weak scaling; n_rays_per_step of config/*.ini), one flat-bucket all-reduce + one fused AMSGrad kernel per
step, blobs re-packed on the device.  --model nerfactor_microfacet | nerfactor (xyz_jitter_std = 0.01, 512 lights)
| shape | nerf (64 + 128 samples, perturb on).

    python scripts/bench_train.py [--model M] [--steps K]            (torchrun for N > 1)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--rays', type=int, default=1024)
    ap.add_argument('--graph', action='store_true', help='capture the step in a hipGraph (optim.GraphedTrainStep)')
    ap.add_argument('--no-update', action='store_true',
                    help='forward + loss + backward only (kernel experiments whose gradients are garbage must not '
                         'reach the weights: operand data changes the clock)')
    ap.add_argument('--precision', default='bf16', choices=['bf16', 'fp32'],
                    help="fp32: grad_precision = fp32, every network forward and backward on the fp32 runtime-shaped kernels")
    ap.add_argument('--fp32-matrix', default='pairs', choices=['pairs', 'native'],
                    help="precision = fp32: bf16 hi / lo operand pairs (default, round 5) or the native fp32 matrix instruction")
    ap.add_argument('--model', default='nerfactor_microfacet',
                    choices=['nerfactor_microfacet', 'nerfactor', 'shape', 'nerf'])
    args = ap.parse_args()
    from nerfactor_amd import build
    build.build()
    from nerfactor_amd import dist as nfx_dist, optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    rank, world = nfx_dist.init_from_env(backend='nccl', device=dev)
    torch.manual_seed(5)  # identical initial weights on every rank (MirroredStrategy semantics)
    extra = dict(shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='') if 'nerfactor' in args.model else {}
    cfg = make_config(args.model, precision=args.precision, fp32_matrix=args.fp32_matrix, **extra)
    model = get_model_class(args.model)(cfg).to(dev)
    opt = optim.make_optimizer(model, cfg)
    rng = np.random.default_rng(100 + rank)
    n = args.rays
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    xyz = t(rng.uniform(-1, 1, size=(n, 3)))
    nrm = torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1)
    cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
    batch = (None, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))), torch.ones(n, 1, device=dev), xyz,
             nrm, t(rng.uniform(size=(n, 512))))
    from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
    mark_all_foreground(batch[5])   # as datasets/nerf_shape.py tags its training batches (rays drawn from alpha > 0.9)
    if args.model == 'nerf':   # rays from the camera towards the unit cube
        batch = (None, None, cam, xyz - cam, t(rng.uniform(size=(n, 3))))
    global_bs = n * world
    step = optim.GraphedTrainStep(model, opt, global_bs) if args.graph else (
        lambda b: optim.train_step(model, b, opt, global_bs))
    if args.no_update:
        def step(b):
            opt.zero_grad()
            pred, gt, kw, _ = model(b, mode='train')
            kw['keep_batch'] = True
            loss = model.compute_loss(pred, gt, **kw).sum() / global_bs
            loss.backward()
            return loss.detach(), None
    for _ in range(args.warmup + (3 if args.graph else 0)):
        step(batch)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss, _ = step(batch)
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    dt = nfx_dist.max_over_ranks(time.perf_counter() - t0, device=dev) / args.steps
    if hasattr(model, 'flush_numerics'):
        model.flush_numerics(block=True)      # raises FloatingPointError if any step's check_numerics failed
    if not np.isfinite(float(loss)):
        raise SystemExit("bench_train: non-finite loss %r after the timed steps — the timing means nothing" % float(loss))
    if rank == 0:
        rows = n * 512 * 2  # clean + jittered visibility rows
        if args.model == 'nerf':   # forward + recomputed forward + dgrad + wgrad of (64 + 192) points per ray
            flops, what = 4 * n * 256 * 1186816, "64+128 samples, perturb on"
        else:
            flops, what = 3 * 2 * (rows * 72320 + 2 * 3 * n * 65664), "512 lights, jitter on"
        print(json.dumps({
            "workload": "%s train step, %d rays/GPU (weak), %s%s, precision = %s" % (
                args.model, n, what, ", hipGraph" if args.graph else "", args.precision + ("" if args.precision == 'bf16' else " (%s)" % args.fp32_matrix)),
            "n_gpus": world, "ms_per_step": dt * 1e3, "rays_per_s": n * world / dt,
            "mlp_flops_per_step_per_gpu": flops, "mlp_tflops": flops / dt / 1e12,
            "final_loss": float(loss)}))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
