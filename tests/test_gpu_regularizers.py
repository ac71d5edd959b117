"""csrc/regularizers.hip against the reference's formulas evaluated by torch in float64: tf.linalg.l2_normalize over rows
with its gradient (nerfactor/util/math.py:63-64 of the reference; nerfactor.py:205-206, 266-270) and the light probe's
smoothness penalties with theirs (nerfactor.py:526-539)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _l2n64(x, eps):
    return x * torch.rsqrt(torch.clamp(torch.sum(x * x, dim=1, keepdim=True), min=eps))


@pytest.mark.parametrize("d", [3, 1, 5, 16])
def test_l2_normalize_rows_and_gradient(nfx_lib, cuda, d):
    from nerfactor_amd import autograd, ops
    rng = np.random.default_rng(d)
    x = rng.normal(size=(3001, d))
    x[:7] *= 1e-4            # inside the epsilon clamp: y = x / sqrt(eps), the norm passes no gradient
    x[7] = 0.
    x[8:12] *= 1e3
    dy = rng.normal(size=x.shape)
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    yr = _l2n64(xr, 1e-6)
    (gr,) = torch.autograd.grad(yr, xr, torch.tensor(dy, dtype=torch.float64))
    xd = torch.tensor(x, dtype=torch.float32, device=cuda, requires_grad=True)
    yd = autograd.l2_normalize(xd, 1e-6)
    assert yd.grad_fn is not None and type(yd.grad_fn).__name__.startswith('L2NormalizeRows')
    (gd,) = torch.autograd.grad(yd, xd, torch.tensor(dy, dtype=torch.float32, device=cuda))
    # fp32 against float64: a few ulp of the result's scale (|y| <= 1; |dx| <= |dy| / |x|)
    assert np.abs(yd.detach().cpu().numpy() - yr.detach().numpy()).max() < 2e-6
    # (scale of a row's gradient: |dy| / max(|x|, sqrt(eps)) — for d = 1 the exact gradient is 0 outside the clamp and what is
    #  left in fp32 is the cancellation of two terms of that size)
    scale = np.abs(dy).max(axis=1, keepdims=True) / np.maximum(np.linalg.norm(x, axis=1, keepdims=True), 1e-3)
    assert (np.abs(gd.cpu().numpy() - gr.numpy()) / scale).max() < 2e-5
    with torch.no_grad():
        assert torch.equal(autograd.l2_normalize(xd), ops.l2_normalize_rows(xd.detach()))
    if d == 3:               # the render path's kernel computes the same thing
        assert torch.equal(ops.l2_normalize_rows(xd.detach()), ops.l2_normalize3(xd.detach(), 1e-6))
    with pytest.raises(Exception, match='1 <= d <= 16'):
        ops.l2_normalize_rows(torch.zeros(4, 17, device=cuda))
    assert ops.l2_normalize_rows(torch.zeros(0, 3, device=cuda)).shape == (0, 3)


@pytest.mark.parametrize("h,w,tv,achro", [(16, 32, 5e-6, 0.), (16, 32, 5e-6, 1e-3), (3, 5, 0.25, 0.5), (1, 1, 1., 1.)])
def test_light_smoothness_and_gradient(nfx_lib, cuda, h, w, tv, achro):
    from nerfactor_amd import autograd
    rng = np.random.default_rng(h * w)
    light = rng.uniform(0., 3., size=(h, w, 3))
    lr = torch.tensor(light, dtype=torch.float64, requires_grad=True)
    dx, dy, dc = lr - torch.roll(lr, 1, 1), lr - torch.roll(lr, 1, 0), lr - torch.roll(lr, 1, 2)
    ref = tv * (dx ** 2 + dy ** 2).sum() + achro * (dc ** 2).sum()
    (gr,) = torch.autograd.grad(ref * 0.7, lr)
    ld = torch.tensor(light, dtype=torch.float32, device=cuda, requires_grad=True)
    got = autograd.LightSmoothness.apply(ld, tv, achro)
    assert got.dim() == 0
    (gd,) = torch.autograd.grad(got * 0.7, ld)
    assert abs(float(got) - float(ref)) <= 2e-6 * abs(float(ref)) + 1e-12
    assert np.abs(gd.cpu().numpy() - gr.numpy()).max() <= 2e-6 * np.abs(gr.numpy()).max() + 1e-12
    again = autograd.LightSmoothness.apply(ld, tv, achro)
    assert torch.equal(got, again)       # one block, fixed reduction order
