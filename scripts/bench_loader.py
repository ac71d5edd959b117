#!/usr/bin/env python
"""End-to-end training epochs THROUGH THE DATASET (datasets/nerf_shape.py -> optim.train_step): what the read-ahead of
datasets/base.py buys.  Synthetic views of the shipped size (512 x 512 pixels x 512 light visibilities = 0.5 GB per
view, cached on the host as trainvali does), 1024 rays per step, `--views` steps per epoch; each configuration
(prefetch 0 | 2) x (eager | hipGraph) runs `--epochs` epochs after one warm-up epoch and checks that the loss sequence
does not depend on the read-ahead.

    python scripts/bench_loader.py [--imh 512] [--views 40] [--epochs 3]
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
    ap.add_argument('--imh', type=int, default=512)
    ap.add_argument('--views', type=int, default=40)
    ap.add_argument('--epochs', type=int, default=3)
    ap.add_argument('--model', default='nerfactor_microfacet')
    ap.add_argument('--only', default='', help="e.g. 'hipGraph:0' = that configuration alone")
    args = ap.parse_args()
    from nerfactor_amd import build
    build.build()
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.datasets.nerf_shape import Dataset as ShapeDataset
    from nerfactor_amd.nerfactor.models import get_model_class
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    h = args.imh
    rng = np.random.default_rng(0)
    t0 = time.perf_counter()
    # three distinct views in host memory, `--views` names cycling over them
    arrays = []
    for v in range(3):
        xyz = rng.uniform(-1, 1, size=(h, h, 3)).astype(np.float32)
        nrm = rng.normal(size=(h, h, 3)).astype(np.float32)
        nrm /= np.linalg.norm(nrm, axis=2, keepdims=True)
        alpha = (rng.uniform(size=(h, h)) < 0.6).astype(np.float32)
        lvis = rng.random(size=(h, h, 512), dtype=np.float32)
        cam = np.broadcast_to(np.float32([2.2, -2.4, 1.7]), (h, h, 3)).copy()
        arrays.append((cam, xyz - cam, rng.random(size=(h, h, 3), dtype=np.float32), alpha, xyz, nrm, lvis))
    gen_s = time.perf_counter() - t0

    class Synth(ShapeDataset):
        def _glob(self):
            return ['view_%03d' % i for i in range(args.views)]

        def _process_example_precache(self, name):
            return (name,) + arrays[int(name[-3:]) % 3]

    results, losses = {}, {}
    for graph in (False, True):
        for prefetch in (0, 2):
            if args.only and args.only != '%s:%d' % ('hipGraph' if graph else 'eager', prefetch):
                continue
            torch.manual_seed(5)
            cfg = make_config(args.model, shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='',
                              n_rays_per_step=1024, prefetch=prefetch)
            model = get_model_class(args.model)(cfg).to(dev)
            opt = optim.make_optimizer(model, cfg)
            ds = Synth(cfg, 'train', device=dev)
            step = optim.GraphedTrainStep(model, opt, 1024) if graph else (
                lambda b: optim.train_step(model, b, opt, 1024))
            seq = []
            for epoch in range(args.epochs + 1):
                if epoch == 1:
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                for batch in ds.build_pipeline(no_batch=True, seed=epoch):
                    loss, _ = step(batch)
                    seq.append(loss.clone())
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / (args.epochs * args.views)
            model.flush_numerics(block=True)
            key = '%s, prefetch %d' % ('hipGraph' if graph else 'eager', prefetch)
            results[key] = dt * 1e3
            print('%s: %.3f ms per step' % (key, dt * 1e3), file=sys.stderr, flush=True)
            losses[key] = torch.stack(seq).cpu().numpy()
    same = {m: bool(np.array_equal(losses[m + ', prefetch 0'], losses[m + ', prefetch 2']))
            for m in ('eager', 'hipGraph') if m + ', prefetch 0' in losses and m + ', prefetch 2' in losses}
    print(json.dumps({
        "workload": "%s training epochs through datasets/nerf_shape.py: %d views of %d x %d x 512 lights cached on the "
                    "host, 1024 rays per step" % (args.model, args.views, h, h),
        "ms_per_step_including_the_loader": results, "loss_sequence_independent_of_prefetch": same,
        "final_loss": {k: float(v[-1]) for k, v in losses.items()}, "host_data_generation_s": gen_s}))


if __name__ == '__main__':
    main()
