#!/usr/bin/env python
"""Where does the hipGraph-captured training step leave the eager one?  (100+ replays of the
NeRFactor steps ended in NaN / 2e5 while the six-step, jitter-off test was bit-identical.)

Runs the same model twice from the same seed — optim.train_step and optim.GraphedTrainStep — keeps the flat parameter
buffer after every step and prints the first step at which the two differ, which parameter differs most there, and
the loss curves.

    python scripts/diag_graph_diverge.py --model nerfactor_microfacet --steps 120 [--jitter 0.01] [--rays 1024]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model', default='nerfactor_microfacet')
    ap.add_argument('--steps', type=int, default=120)
    ap.add_argument('--rays', type=int, default=1024)
    ap.add_argument('--jitter', default='0.01')
    ap.add_argument('--seed', type=int, default=11)
    ap.add_argument('--data-seed', type=int, default=7)
    ap.add_argument('--batches', type=int, default=8)
    ap.add_argument('--keep-vis', action='store_true', help='keep the to_vis of steps 3 and 4 alive, as trainvali does')
    ap.add_argument('--ids', action='store_true', help='batch[0] = list of view ids (as the datasets produce)')
    ap.add_argument('--same-batch', action='store_true', help='feed one batch every step (what bench_train.py does)')
    args = ap.parse_args()
    from nerfactor_amd import build
    build.build()
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
    from nerfactor_amd.nerfactor.models import get_model_class
    dev = torch.device('cuda', 0)
    n = args.rays
    extra = dict(shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='') if 'nerfactor' in args.model else {}

    def batches():
        rng = np.random.default_rng(args.data_seed)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        out = []
        for bi in range(1 if args.same_batch else args.batches):
            xyz = t(rng.uniform(-1, 1, size=(n, 3)))
            nrm = torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1)
            cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
            out.append(((['view%d' % bi] * n) if args.ids else None, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))),
                        mark_all_foreground(torch.ones(n, 1, device=dev)), xyz, nrm, t(rng.uniform(size=(n, 512)))))
        return out

    def run(graph):
        torch.manual_seed(args.seed)
        cfg = make_config(args.model, xyz_jitter_std=args.jitter, **extra)
        model = get_model_class(args.model)(cfg).to(dev)
        opt = optim.make_optimizer(model, cfg)
        step = optim.GraphedTrainStep(model, opt, n, warmup=2) if graph else (
            lambda b: optim.train_step(model, b, opt, n))
        bs = batches()
        losses, flats, grads, kept = [], [], [], []
        for i in range(args.steps):
            loss, to_vis = step(bs[i % len(bs)])
            losses.append(loss.clone())
            if args.keep_vis and i in (3, 4):
                kept.append(to_vis)
            flats.append(opt.flat.clone())
            grads.append(opt.bucket.flat.clone())
        torch.cuda.synchronize()
        names, off = [], 0
        ids = {id(p): k for k, p in model.named_parameters()}
        for p in opt.params:
            names.append((ids.get(id(p), '?'), off, off + p.numel()))
            off += p.numel()
        return torch.stack(losses).cpu().numpy(), flats, grads, names

    l0, f0, g0, names = run(False)
    l1, f1, g1, _ = run(True)
    first = next((i for i in range(args.steps) if not torch.equal(f0[i], f1[i])), None)
    firstg = next((i for i in range(args.steps) if not torch.equal(g0[i], g1[i])), None)
    rep = {'model': args.model, 'jitter': args.jitter, 'rays': n, 'steps': args.steps, 'same_batch': args.same_batch,
           'first_param_diff_step': first, 'first_grad_diff_step': firstg,
           'loss_eager_head': l0[:8].tolist(), 'loss_graph_head': l1[:8].tolist(),
           'loss_eager_tail': l0[-4:].tolist(), 'loss_graph_tail': l1[-4:].tolist(),
           'first_nonfinite_graph': next((i for i in range(args.steps) if not np.isfinite(l1[i])), None),
           'first_nonfinite_eager': next((i for i in range(args.steps) if not np.isfinite(l0[i])), None)}
    if firstg is not None:
        d = (g0[firstg][:-1] - g1[firstg][:-1]).abs()
        rows = []
        for nm, a, b in names:
            m = float(d[a:b].max())
            if m > 0 or not np.isfinite(m):
                rows.append((nm, m, float(g0[firstg][a:b].abs().max()), float(g1[firstg][a:b].abs().max())))
        rep['grad_diff_at_first'] = sorted(rows, key=lambda r: -r[1] if np.isfinite(r[1]) else -1e30)[:12]
    # how the gap grows
    rep['loss_eager'] = [round(float(x), 5) for x in l0]
    rep['loss_graph_differs_at'] = [i for i in range(args.steps) if not (l0[i] == l1[i] or (np.isnan(l0[i]) and np.isnan(l1[i])))][:10]
    rep['max_param_gap_every_10'] = [float((f0[i] - f1[i]).abs().max()) for i in range(0, args.steps, 10)]
    print(json.dumps(rep))


if __name__ == '__main__':
    main()
