"""Soak of the NeRF backward's two-waves-per-SIMD kernels against their one-wave forms (the regime of the
round-3..6 hazard): `nerf_bwd_ring_kernel<8, LIST>` (default) vs `<4, LIST>` (option nerf_bwd_nw = 4), list form and every-point
form, on fresh sparse upstream gradients every launch — every weight gradient must agree bit for bit.

    python scripts/soak_nerf_bwd.py [--launches 2000] [--rays 768] [--samples 192]"""
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--launches', type=int, default=2000)
    ap.add_argument('--rays', type=int, default=768)
    ap.add_argument('--samples', type=int, default=192)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    ks_np, bs_np = common.nerf_layers(common.nerf_nets(seed=5, opaque=False)[0])
    blob = ops.pack_nerf_train_weights(ks_np, bs_np).to(dev)
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    n, s = args.rays, args.samples
    rayo = (torch.rand((n, 3), device=dev, generator=g) * 2 - 1)
    rayd = torch.nn.functional.normalize(torch.randn((n, 3), device=dev, generator=g), dim=1)
    z = torch.sort(torch.rand((n, s), device=dev, generator=g) * 2.5 + 0.5, 1)[0].contiguous()

    def grads(nw, rows, d_rgbs):
        _capi.set_option('nerf_bwd_nw', nw)
        _capi.set_option('nerf_bwd_rows', rows)
        dks = [torch.zeros(k.shape, device=dev) for k in ks_np]
        dbs = [torch.zeros(b.shape, device=dev) for b in bs_np]
        ops.nerf_mlp_bwd(rayo, rayd, z, d_rgbs, blob, dks, dbs)
        return torch.cat([t.reshape(-1) for t in dks + dbs])

    bad = {0: 0, 1: 0}
    rows_done = {0: 0, 1: 0}
    t0 = time.time()
    for it in range(args.launches):
        keep = 0.05 + 0.9 * ((it * 37) % 100) / 100.
        d = torch.randn((n, s, 4), device=dev, generator=g)
        d = d * (torch.rand((n, s, 1), device=dev, generator=g) < keep)
        for rows in (1, 0):
            a, b = grads(8, rows, d), grads(4, rows, d)
            if not torch.equal(a, b):
                bad[rows] += 1
                print("launch %d rows=%d: %d gradient words differ" % (it, rows, int((a != b).sum())), flush=True)
            rows_done[rows] += n * s if rows == 0 else int((d != 0).any(-1).sum())
    torch.cuda.synchronize()
    for k in ('nerf_bwd_nw', 'nerf_bwd_rows'):
        _capi.unset_option(k)
    print(json.dumps({"launches_per_form": args.launches, "points_per_launch": n * s,
                      "rows_through_the_list_form": rows_done[1], "rows_through_the_every_point_form": rows_done[0],
                      "launches_with_a_difference": {"list": bad[1], "every_point": bad[0]},
                      "seconds": round(time.time() - t0, 1)}))
    sys.exit(1 if bad[0] or bad[1] else 0)


if __name__ == '__main__':
    main()
