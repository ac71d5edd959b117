"""The differentiable local-frame / Rusinkiewicz geometry of the plugin (nerfactor/util/geom.py, util/math.py: what a
training call at grad_precision = fp32 differentiates instead of the fused shading kernels) against the oracle's
restatement of the reference (oracle/torch_train_ref.py, itself pinned to the reference's own training step): values and
gradients in float64, the degenerate configurations the custom gradients exist for included (view = light direction:
theta_d = 0; a half vector on the pole: atan2(0, 0))."""
import math

import numpy as np
import torch

from nerfactor_amd.nerfactor.util import geom, math as mathutil
from oracle import torch_train_ref as T


def _oracle_rusink(lf, vf):
    a, bb = T.l2n(lf, 1, 1e-6), T.l2n(vf, 1, 1e-6)
    h = T.l2n((a + bb) / 2, 1, 1e-6)
    theta_h = T.SafeAcos.apply(h[:, 2])
    phi_h = T.SafeAtan2.apply(h[:, 1], h[:, 0])
    diff = T._rot_vec(T._rot_vec(bb, (0., 0., 1.), -phi_h), (0., 1., 0.), -theta_h)
    theta_d = T.SafeAcos.apply(diff[:, 2])
    phi_d = torch.remainder(T.SafeAtan2.apply(diff[:, 1], diff[:, 0]), math.pi)
    return torch.stack((phi_d, theta_h, theta_d), 1)


def test_rusinkiewicz_coordinates_and_their_custom_gradients():
    rng = np.random.default_rng(0)
    l = rng.normal(size=(400, 3))
    v = rng.normal(size=(400, 3))
    v[:20] = l[:20]                                  # theta_d = 0: phi_d meaningless, its gradient must stay finite
    l[20:30] = (0., 0., 2.)
    v[20:30] = (0., 0., .5)                          # half vector on the pole: atan2(0, 0)
    w = rng.normal(size=(400, 3))
    outs = []
    for fn in (geom.dir2rusink_autograd, _oracle_rusink):
        a = torch.tensor(l, dtype=torch.float64, requires_grad=True)
        b = torch.tensor(v, dtype=torch.float64, requires_grad=True)
        r = fn(a, b)
        (r * torch.tensor(w)).sum().backward()
        outs.append((r.detach().numpy(), a.grad.numpy(), b.grad.numpy()))
    for got, want in zip(*outs):
        assert np.isfinite(got).all()
        np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-12)
    r = outs[0][0]
    assert (r[:, 0] >= 0).all() and (r[:, 0] < math.pi + 1e-12).all() and (r[:, 1:] >= 0).all()


def test_safe_acos_and_atan2_slopes_at_the_singular_points():
    x = torch.tensor([-1.5, -1., 0., 1., 2.], dtype=torch.float64, requires_grad=True)
    y = mathutil.safe_acos(x)
    y.sum().backward()
    assert torch.allclose(y.detach(), torch.acos(torch.clamp(x.detach(), -1., 1.)))
    xc = torch.clamp(x.detach(), -1., 1.)
    assert torch.allclose(x.grad, -1. / (torch.sqrt(1. - xc ** 2 + 1e-6) + 1e-6))     # finite at +-1, and outside
    a = torch.zeros(3, dtype=torch.float64, requires_grad=True)
    b = torch.zeros(3, dtype=torch.float64, requires_grad=True)
    mathutil.safe_atan2(a, b).sum().backward()
    assert bool((a.grad == 0).all()) and bool((b.grad == 0).all())                     # 0 / eps, not NaN


def test_world_to_local_frames_match_the_oracle():
    rng = np.random.default_rng(1)
    n = torch.tensor(rng.normal(size=(64, 3)), dtype=torch.float64)
    n[0] = torch.tensor((0., 0., 1.))                # colinear with the up vector but for the eps offset
    rot = geom.gen_world2local(n)
    nn = T.l2n(n, 1, 1e-6)
    up = (torch.tensor((0., 0., 1.), dtype=n.dtype) + 1e-6).expand_as(nn)
    t = T.l2n(torch.cross(nn, up, dim=1), 1, 1e-6)
    b = T.l2n(torch.cross(nn, t, dim=1), 1, 1e-6)
    torch.testing.assert_close(rot, torch.stack((t, b, nn), 1), rtol=1e-12, atol=1e-12)
