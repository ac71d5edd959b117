"""CPU restatement of nerfactor/geometry_from_nerf.py (TEST INFRASTRUCTURE, see oracle/__init__.py; pinned to the
reference's own compute_depth_and_normal / compute_light_visibility run on the NumPy TF shim with a forward-mode
GradientTape, tests/golden/make_reference_golden.py): expected depth / normal from a trained NeRF and light visibility by shadow-ray marching.
NumPy for the marching (oracle/nerf_ref.py), torch-CPU autograd only for d sigma / dx — the reference takes it with
GradientTape.batch_jacobian (geometry_from_nerf.py:289-297).

  compute_depth_and_normal   geometry_from_nerf.py:249-319
  compute_light_visibility   geometry_from_nerf.py:177-246
  eval_sigma_mlp             geometry_from_nerf.py:322-350   (relu on sigma_out(enc(embed(x))))
"""
import numpy as np
import torch

from . import nerf_ref, torch_ref


def in_bounds(pts, bbox):
    """check_bounds (geometry_from_nerf.py:365-378): bbox = (x_min, x_max, y_min, y_max, z_min, z_max) or None."""
    if bbox is None:
        return np.ones(pts.shape[:-1], bool)
    lo, hi = np.asarray(bbox[0::2], pts.dtype), np.asarray(bbox[1::2], pts.dtype)
    return np.all((pts >= lo) & (pts <= hi), -1)


def eval_sigma(pts, net, dtype=np.float32, bbox=None):
    """relu(sigma)[N,S] at pts[N,S,3] — eval_sigma_mlp: 0 outside the scene bounding box."""
    depth = len(net['enc'])
    pe = nerf_ref.embed(pts.reshape(-1, 3).astype(dtype), 10)
    feat = nerf_ref.mlp(pe, net['enc'], ['relu'] * depth, skip_at=[depth // 2])
    sigma = nerf_ref.mlp(feat, net['sigma_out'], [None])
    return np.maximum(sigma, 0.).reshape(pts.shape[:2]) * in_bounds(pts, bbox)


def sigma_and_normal(pts, net):
    """(relu(sigma)[N,S], -l2_normalize(d relu(sigma) / dx)[N,S,3]) in float64 (:280-297)."""
    tnet = {k: [(w.double(), b.double()) for w, b in v] for k, v in torch_ref.to_torch_net(net).items()}
    x = torch.tensor(pts.reshape(-1, 3), dtype=torch.float64, requires_grad=True)
    depth = len(tnet['enc'])
    feat = torch_ref.mlp(torch_ref.embed(x, 10), tnet['enc'], ['relu'] * depth, skip_at=[depth // 2])
    sigma = torch.relu(torch_ref.mlp(feat, tnet['sigma_out'], [None]))
    (jac,) = torch.autograd.grad(sigma.sum(), x)        # rows are independent: the batch Jacobian
    jac = jac.numpy()
    normal = -jac / np.sqrt(np.maximum((jac ** 2).sum(-1, keepdims=True), 1e-12))   # tf.linalg.l2_normalize
    return sigma.detach().numpy().reshape(pts.shape[:2]), normal.reshape(pts.shape)


def _march(o, d, net_coarse, net_fine, near, far, n_coarse, n_fine, want_normal, bbox=None):
    z = nerf_ref.gen_z(near, far, n_coarse, o.shape[0])
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    w = nerf_ref.accumulate_sigma(eval_sigma(pts, net_coarse, bbox=bbox), z, d)
    z = nerf_ref.gen_z_fine(z, w, n_fine)
    pts = o[:, None, :] + d[:, None, :] * z[:, :, None]
    if want_normal:
        sigma, normal = sigma_and_normal(pts, net_fine)
        sigma = sigma * in_bounds(pts, bbox)           # :303-305: sigma overridden, the normal is not
    else:
        sigma, normal = eval_sigma(pts, net_fine, bbox=bbox), None
    w = nerf_ref.accumulate_sigma(sigma.astype(np.float32), z, d)
    occu, depth = w.sum(-1), (w * z).sum(-1)
    exp_normal = (w[:, :, None] * normal).sum(1) if want_normal else None
    return occu, depth, exp_normal


def compute_depth_and_normal(rayo, rayd, net_coarse, net_fine, near=2., far=6., n_samples_coarse=64,
                             n_samples_fine=128, bbox=None):
    """(occu, exp_depth, exp_normal): 64 + n_samples_coarse coarse samples, + 64 + n_samples_fine importance samples,
    all evaluated with the fine network."""
    return _march(rayo, rayd, net_coarse, net_fine, near, far, 64 + n_samples_coarse, 64 + n_samples_fine, True,
                  bbox)


def compute_light_visibility(surf, normal, lxyz, net_coarse, net_fine, lvis_far=1., lvis_near=.1, n_samples_coarse=64,
                             n_samples_fine=128, bbox=None):
    """lvis[n, L] = 1 - occupancy along the ray to every front-lit light, 0 for back-lit ones."""
    n, n_lights = surf.shape[0], lxyz.shape[0]
    surf2l = lxyz[None] - surf[:, None]
    surf2l = surf2l / np.sqrt(np.maximum((surf2l ** 2).sum(-1, keepdims=True), 1e-12))
    front = (surf2l * normal[:, None]).sum(-1) > 0
    lvis = np.zeros((n, n_lights), np.float32)
    if front.any():
        o = np.broadcast_to(surf[:, None], surf2l.shape)[front].astype(np.float32)
        occu, _, _ = _march(o, surf2l[front].astype(np.float32), net_coarse, net_fine, lvis_near, lvis_far,
                            64 + n_samples_coarse, 64 + n_samples_fine, False, bbox)
        lvis[front] = 1. - occu
    return lvis
