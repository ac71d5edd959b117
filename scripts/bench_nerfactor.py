#!/usr/bin/env python
"""Secondary benchmark (BASELINE.json configs[2] and [4], SURVEY.md §8d "C3"/"C5"): NeRFactor full
render of one 800x800 view (60 % foreground), 512 lights, trained light + 8 novel probes, through the
model plugin.  Prints one JSON line; the driver contract (`bench.py`) stays the NeRF render.

    python scripts/bench_nerfactor.py [--model nerfactor_microfacet|nerfactor] [--steps K]
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
    ap.add_argument('--model', default='nerfactor_microfacet')
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--n', type=int, default=640000)
    ap.add_argument('--probes', type=int, default=8)
    args = ap.parse_args()
    from nerfactor_amd import build
    build.build()
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    dev = torch.device('cuda:0')
    torch.manual_seed(5)
    cfg = make_config(args.model, shape_mode='finetune', shape_model_ckpt='none', brdf_model_ckpt='none',
                      test_envmap_dir='', xyz_jitter_std='0')
    model = get_model_class(args.model)(cfg).to(dev)
    rng = np.random.default_rng(1)
    for i in range(args.probes):
        model.add_probe('p%d' % i, np.exp(rng.normal(size=(16, 32, 3))).astype(np.float32))
    n = args.n
    xyz = torch.from_numpy(rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)).to(dev)
    nrm = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
    alpha = (torch.rand(n, 1, device=dev) < 0.6).float()
    cam = torch.tensor([2.2, -2.4, 1.7], device=dev).expand(n, 3).contiguous()
    batch = (None, None, cam, torch.zeros(n, 3, device=dev), torch.rand(n, 3, device=dev), alpha, xyz, nrm,
             torch.rand(n, 512, device=dev))
    n_fg = int(alpha.sum().item())

    def step():
        return model(batch, mode='test', relight_probes=True)[0]['rgb_probes']

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    # kernel-level: light visibility alone
    from nerfactor_amd import _capi, ops
    blob = model._blob128('lvis_mlp', 'lvis_out', _capi.IN_XYZ_LDIR, 1)
    xm = xyz[alpha[:, 0] > 0].contiguous()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ops.lvis_fwd(xm, model.lxyz.reshape(-1, 3), blob)
    ev[0].record()
    for _ in range(3):
        ops.lvis_fwd(xm, model.lxyz.reshape(-1, 3), blob)
    ev[1].record()
    torch.cuda.synchronize()
    lvis_s = ev[0].elapsed_time(ev[1]) / 3 * 1e-3
    rows = n_fg * 512
    print(json.dumps({
        "workload": "%s full render, %d surface points (%d foreground), 512 lights, 1+%d lights" % (
            args.model, n, n_fg, args.probes),
        "points_per_s": n / dt, "foreground_points_per_s": n_fg / dt, "ms_per_view": dt * 1e3,
        "lvis_kernel_ms": lvis_s * 1e3,
        "lvis_algorithmic_tflops": rows * 2 * 72320 / lvis_s / 1e12,
        "lvis_executed_tflops": rows * 2 * 61440 / lvis_s / 1e12,
        "finite": bool(torch.isfinite(out).all().item())}))


if __name__ == '__main__':
    main()
