"""Soak of the density-only kernel in the render kernel's dataflow (csrc/nerf_sigma_v6.hip, option sigma_variant = 1) against
nerf_sigma_geo_kernel (sigma_variant = 0): fresh sample positions every launch, every density must agree bit for bit.

    python scripts/soak_sigma_v6.py [--launches 2000] [--rays 4096] [--samples 320]"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfactor_amd import _capi, ops  # noqa: E402
from tests import common  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--launches', type=int, default=2000)
    ap.add_argument('--rays', type=int, default=4096)
    ap.add_argument('--samples', type=int, default=320)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    blobs = [ops.pack_nerf_geom_weights(*common.nerf_layers(net)).to(dev)
             for net in (gi.trained_nerf_nets()[1], common.nerf_nets(seed=7)[1])]
    g = torch.Generator(device=dev)
    g.manual_seed(2)
    n, s = args.rays, args.samples
    bad, t0 = 0, time.time()
    for it in range(args.launches):
        rayo = torch.rand((n, 3), device=dev, generator=g) * 4 - 2
        rayd = torch.nn.functional.normalize(torch.randn((n, 3), device=dev, generator=g), dim=1)
        z = torch.sort(torch.rand((n, s), device=dev, generator=g) * 3, 1)[0].contiguous()
        blob = blobs[it & 1]
        _capi.set_option('sigma_variant', 0)
        a = ops.nerf_sigma_fwd(rayo, rayd, z, blob)
        _capi.set_option('sigma_variant', 1)
        b = ops.nerf_sigma_fwd(rayo, rayd, z, blob)
        if not torch.equal(a, b):
            bad += 1
            print("launch %d: %d densities differ" % (it, int((a != b).sum())), flush=True)
    torch.cuda.synchronize()
    _capi.unset_option('sigma_variant')
    print(json.dumps({"launches": args.launches, "points_per_launch": n * s, "points": args.launches * n * s,
                      "launches_with_a_difference": bad, "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if bad else 0)


if __name__ == '__main__':
    main()
