"""Recipe of tests/golden/nerf_trained_fp16.npz: "trained-like" weights for the NeRF parity fixtures.

    python tests/golden/make_trained_nerf_weights.py [--steps 700]          (CPU, ~25 min on 8 cores)

The random-weight fixtures leave up to 30 % of the rays undecided at the reference formula's discontinuity (the last
sample of a ray has dist = 1e10, so alpha_last = [sigma_last > 0] exactly, and a glorot network puts sigma_last within
bf16 noise of 0 on many rays).  A network TRAINED on a scene puts empty space at a robustly negative density, so this
script fits the coarse + fine networks (models/nerf.py:53-71 shapes) to an analytic scene — the unit sphere of
tests/synth_scene.py, white background, cameras on the radius-4 orbit, near 2 / far 6 — with the oracle's training loss
(oracle/torch_train_ref.py:nerf_loss, pinned to the reference's) and torch Adam, fp32 on the CPU, seed 0.
The weights are stored as float16 (training is not bit-reproducible across BLAS builds, so the recipe alone would not
reproduce the committed reference outputs): 2 x 595 844 parameters, 2.3 MiB."""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import nerf_ref, torch_train_ref as T  # noqa: E402
from tests import common  # noqa: E402

ANGLE_X = 0.6911


def scene_rays(rng, n):
    """n random camera rays of random views + the analytic colour of the shaded unit sphere on white."""
    v = rng.normal(size=(n, 3))
    v[:, 2] = np.abs(v[:, 2]) + 0.3
    cam = 4. * v / np.linalg.norm(v, axis=1, keepdims=True)
    rayo, rayd = np.empty((n, 3)), np.empty((n, 3))
    for i in range(n):   # one random pixel direction per random camera
        c2w = nerf_ref.lookat_cam_to_world(cam[i])
        fl = .5 / np.tan(.5 * ANGLE_X)
        px = rng.uniform(-.5, .5, 2)
        rayd[i] = c2w[:3, :3] @ np.array([px[0] / fl, -px[1] / fl, -1.])
        rayo[i] = cam[i]
    d = rayd / np.linalg.norm(rayd, axis=1, keepdims=True)
    b = (rayo * d).sum(1)
    disc = b * b - ((rayo * rayo).sum(1) - 1.)
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.))
    xyz = rayo + t[:, None] * d
    albedo = 0.5 + 0.4 * np.sin(3. * xyz)
    light = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
    shade = 0.25 + 0.75 * np.clip((xyz * light).sum(1, keepdims=True), 0, None)
    rgb = np.where(hit[:, None], np.clip(albedo * shade, 0, 1), 1.)
    return rayo.astype(np.float32), rayd.astype(np.float32), rgb.astype(np.float32)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=700)
    ap.add_argument('--rays', type=int, default=384)
    args = ap.parse_args()
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    P = {}
    for pref, net in zip(('coarse_', 'fine_'), common.nerf_nets(seed=0, opaque=False, random_bias=False)):
        for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
            for i, (k, b) in enumerate(net[part]):
                P['net_%s%s_layer%d.kernel' % (pref, part, i)] = torch.tensor(k, requires_grad=True)
                P['net_%s%s_layer%d.bias' % (pref, part, i)] = torch.tensor(b, requires_grad=True)
    opt = torch.optim.Adam(list(P.values()), lr=5e-4)
    t0 = time.time()
    for step in range(args.steps):
        rayo, rayd, rgb = (torch.from_numpy(a) for a in scene_rays(rng, args.rays))
        n, nc, nf = rayo.shape[0], 32, 64
        u0, u1 = torch.rand(n, nc), torch.rand(n, nf)
        loss = T.nerf_loss(P, rayo, rayd, rgb, u0, torch.randn(n, nc), u1, torch.randn(n, nc + nf), n_coarse=nc,
                           n_fine=nf, noise_std=1. if step < args.steps // 2 else 0.).mean()
        opt.zero_grad()
        loss.backward()
        opt.step()
        if step % 25 == 0 or step == args.steps - 1:
            print('step %4d  loss %.5f  (%.0f s)' % (step, float(loss), time.time() - t0), flush=True)
    out = {k: v.detach().numpy().astype(np.float16) for k, v in P.items()}
    path = os.path.join(HERE, 'nerf_trained_fp16.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%.1f KiB)' % (path, os.path.getsize(path) / 1024))


if __name__ == '__main__':
    main()
