"""util.geom — local shading frames and Rusinkiewicz coordinates (reference:
nerfactor/util/geom.py:119-192)."""
import torch

from nerfactor_amd import ops

from .math import safe_l2_normalize


def gen_world2local(normal, eps=1e-6):
    """[N,3] normals -> [N,3,3] rotations whose ROWS are (tangent, binormal, normal)."""
    n = safe_l2_normalize(normal, axis=1, eps=eps)
    z = torch.tensor((0., 0., 1.), dtype=n.dtype, device=n.device) + eps  # avoids colinearity
    t = safe_l2_normalize(torch.cross(n, z.expand_as(n), dim=1), axis=1, eps=eps)
    b = safe_l2_normalize(torch.cross(n, t, dim=1), axis=1, eps=eps)
    return torch.stack((t, b, n), dim=1)


def dir2rusink(a, b):
    """(phi_d, theta_h, theta_d) of direction pairs a (light), b (view), both [N,3] in the local
    frame — evaluated by libnfx (nfx_dir2rusink)."""
    return ops.dir2rusink(a, b)
