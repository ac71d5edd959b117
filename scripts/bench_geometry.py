#!/usr/bin/env python
"""Secondary benchmark: geometry_from_nerf on one synthetic view — camera march with density
gradients (128 + 320 samples per ray) and shadow-ray march to 512 lights; 128 coarse + 192 importance samples per ray (the fine pass evaluates all 320).

    python scripts/bench_geometry.py [--imh 256] [--surf 4096]
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
    ap.add_argument('--imh', type=int, default=256)
    ap.add_argument('--surf', type=int, default=4096, help="surface points marched towards the 512 lights")
    args = ap.parse_args()
    from nerfactor_amd import build
    build.build()
    from nerfactor_amd.nerfactor import geometry_from_nerf as G
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from tests import common
    dev = torch.device('cuda', 0)
    torch.manual_seed(0)
    cfg = make_config('nerf')
    model = get_model_class('nerf')(cfg).to(dev)
    with torch.no_grad():
        for pref in ('coarse_', 'fine_'):
            layer = model.net[pref + 'sigma_out'].layers[0]
            layer.kernel.mul_(8.)
            layer.bias.add_(0.5)
    rayo, rayd = common.camera_rays(args.imh, args.imh)
    rayo, rayd = torch.from_numpy(rayo).to(dev), torch.from_numpy(rayd).to(dev)
    rayd = torch.nn.functional.normalize(rayd, dim=1)
    out = {}
    with torch.no_grad():
        for rep in range(2):   # first pass = warm-up (packing, allocator)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            occu, depth, normal = G.compute_depth_and_normal(model, rayo, rayd, cfg)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            surf = (rayo + rayd * depth[:, None])[:args.surf].contiguous()
            nrm = normal[:args.surf].contiguous()
            lvis = G.compute_light_visibility(model, surf, nrm, cfg)
            torch.cuda.synchronize()
            t2 = time.perf_counter()
        n = rayo.shape[0]
        lxyz, _ = G.gen_light_xyz(16, 32)
        lx = torch.as_tensor(lxyz.reshape(-1, 3).astype(np.float32), device=dev)
        s2l = torch.nn.functional.normalize(lx[None] - surf[:, None], dim=2)
        pairs = int(((s2l * nrm[:, None]).sum(-1) > 0).sum())
    enc_flop = 2 * (63 * 256 + 6 * 256 * 256 + 319 * 256 + 256)          # encoder + sigma_out, per density sample
    out = {
        "workload": "geometry_from_nerf, %dx%d view, 128 coarse + 320 fine-net samples per ray" % (args.imh, args.imh),
        "depth_normal_ms": (t1 - t0) * 1e3, "rays_per_s": n / (t1 - t0),
        # coarse forward 128 samples; fine forward + reverse sweep (~1x the forward MACs) on 320 samples
        "depth_normal_tflops": n * (128 + 2 * 320) * enc_flop / (t1 - t0) / 1e12,
        "lvis_surface_points": args.surf, "lvis_front_lit_pairs": pairs, "lvis_ms": (t2 - t1) * 1e3,
        "lvis_pairs_per_s": pairs / (t2 - t1), "lvis_tflops": pairs * (128 + 320) * enc_flop / (t2 - t1) / 1e12,
        "finite": bool(torch.isfinite(normal).all() and torch.isfinite(lvis).all())}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
