"""NeRF training step with the backward over every point (option nerf_bwd_rows = 0) and over the points with a gradient
(default), on the bench's freshly initialised networks and on the networks fitted to a scene (tests/golden): share of the
points with a gradient, step time (one hipGraph replay per step), the backward calls' time.

    python scripts/nerf_bwd_rows.py            (GPU box; prints one JSON object)"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfactor_amd import _capi, ops, optim, synth  # noqa: E402
from nerfactor_amd.nerfactor.config import make_config  # noqa: E402
from nerfactor_amd.nerfactor.models import get_model_class  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402

N_RAYS = 1024
dev = torch.device('cuda:0')


def make_model(nets):
    torch.manual_seed(0)
    model = get_model_class('nerf')(make_config('nerf'))
    if nets is not None:
        with torch.no_grad():
            for pref, net in zip(('coarse_', 'fine_'), nets):
                for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
                    for layer, (k, b) in zip(model.net[pref + part].layers, net[part]):
                        layer.kernel.copy_(torch.from_numpy(np.asarray(k, np.float32)))
                        layer.bias.copy_(torch.from_numpy(np.asarray(b, np.float32)))
    return model.to(dev)


def batch_of(kind):
    rng = np.random.default_rng(7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
    if kind == 'glorot':      # bench.py train_leg: rays from the camera towards the unit cube
        xyz = rng.uniform(-1, 1, size=(N_RAYS, 3))
        cam = np.broadcast_to([2.2, -2.4, 1.7], (N_RAYS, 3))
        return (None, None, t(cam), t(xyz - cam), t(rng.uniform(size=(N_RAYS, 3))))
    rayo, rayd = synth.camera_rays(800, 800, cam_loc=(3.2, -0.1, 2.4))      # the view bench.py renders
    idx = rng.choice(rayo.shape[0], N_RAYS, replace=False)
    return (None, None, t(rayo[idx]), t(rayd[idx]), None)      # (colours: the fitted networks' own render, measure())


def measure(kind, rows):
    _capi.set_option('nerf_bwd_rows', rows)
    model = make_model(None if kind == 'glorot' else gi.trained_nerf_nets())
    opt = optim.make_optimizer(model, model.config)
    batch = batch_of(kind)
    if batch[4] is None:      # a scene in training: the targets are what the scene looks like, the densities stay where they are
        with torch.no_grad():
            rgb = model(batch[:4] + (torch.zeros_like(batch[2]),), mode='test')[0]['fine']
        batch = batch[:4] + (rgb.clamp(0, 1),)

    def count():
        ops.NERF_BWD_STATS = []
        optim.train_step(model, batch, opt, N_RAYS)
        torch.cuda.synchronize()
        st = [(int(c.item()), m) for c, m in ops.NERF_BWD_STATS]
        ops.NERF_BWD_STATS = None
        return st
    stats = count()
    step = optim.GraphedTrainStep(model, opt, N_RAYS)
    for _ in range(8):
        step(batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 100
    for _ in range(n):
        loss, _ = step(batch)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    r = {"ms_per_step": ms, "loss": float(loss), "points": [m for _, m in stats]}
    if rows:      # (the list is only built in this mode)
        after = count()
        r["points_with_gradient"] = [c for c, _ in stats]
        r["points_with_gradient_after_the_steps"] = [c for c, _ in after]
        r["frac"] = sum(c for c, _ in stats) / sum(m for _, m in stats)
        r["frac_after_the_steps"] = sum(c for c, _ in after) / sum(m for _, m in after)
    return r


def main():
    out = {}
    kinds = [a for a in sys.argv[1:] if a in ('glorot', 'fitted')] or ['glorot', 'fitted']
    modes = [int(a) for a in sys.argv[1:] if a in ('0', '1')] or [0, 1, 0, 1]      # (python ... fitted 1: one case, for rocprofv3)
    for kind in kinds:
        out[kind] = {}
        for rows in modes:
            r = measure(kind, rows)
            k = 'rows_with_gradient' if rows else 'every_point'
            if k in out[kind]:
                out[kind][k]["ms_per_step_second_run"] = r["ms_per_step"]
            else:
                out[kind][k] = r
    _capi.unset_option('nerf_bwd_rows')
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
