"""GPU parity, training-side kernels: fused MLP backward (re-computed forward + dgrad chain + wgrad
GEMMs) vs torch autograd of the same network in fp32, and the fused AMSGrad update vs a NumPy
restatement of Keras' Adam(amsgrad=True) (TF 2.2 OptimizerV2 semantics)."""
import numpy as np
import pytest
import torch

from oracle import nerf_ref, nerfactor_ref as R
from tests.test_gpu_nerfactor import dev, net128, scene

pytestmark = pytest.mark.gpu


def torch_embed(x, L):
    parts = [x]
    for k in range(L):
        parts += [torch.sin(x * 2. ** k), torch.cos(x * 2. ** k)]
    return torch.cat(parts, -1)


def q16(t):
    """bf16 rounding with a straight-through gradient: the forward sees the operand rounding of the
    MFMA path (weights and layer inputs in bf16), the backward the same ReLU masks and rounded weights."""
    return t + (t.detach().float().to(torch.bfloat16).to(t.dtype) - t.detach())


def torch_mlp128(x, ks, bs, out_act, quant=False):
    q = q16 if quant else (lambda t: t)
    h = x
    for i in range(4):
        h = torch.relu(q(h) @ q(ks[i]) + bs[i])
        if i == 2:
            h = torch.cat((h, x), -1)
    y = q(h) @ q(ks[4]) + bs[4]
    return {None: lambda v: v, 'sigmoid': torch.sigmoid}[out_act](y)


def _params(layers, out, device):
    ks = [torch.tensor(k, device=device, dtype=torch.float64, requires_grad=True) for k, _ in layers + out]
    bs = [torch.tensor(b, device=device, dtype=torch.float64, requires_grad=True) for _, b in layers + out]
    return ks, bs


def _check_grads(got, want, what, tol=4e-2):
    errs = []
    for i, (g, w) in enumerate(zip(got, want)):
        g, w = g.double().cpu().numpy(), w.cpu().numpy()
        scale = np.abs(w).max() + 1e-12
        d = np.abs(g - w)
        # max error relative to the largest entry, and the relative Frobenius error
        errs.append((d.max() / scale, np.linalg.norm(d) / (np.linalg.norm(w) + 1e-12),
                     tuple(int(v) for v in np.unravel_index(np.argmax(d), d.shape))))
    assert all(e[1] < tol for e in errs), "%s: (max/scale, frobenius rel, argmax) per layer = %s" % (what, errs)


@pytest.mark.parametrize("out_dim,act,scale,n", [(3, 'sigmoid', .77, 1000), (3, None, 1., 77), (1, 'sigmoid', 1., 300)])
def test_mlp128_xyz_backward_vs_autograd(nfx_lib, cuda, out_dim, act, scale, n):
    from nerfactor_amd import ops
    layers, out = net128(80 + out_dim, 63, out_dim)
    rng = np.random.default_rng(81)
    xyz = rng.uniform(-1.2, 1.2, size=(n, 3)).astype(np.float32)
    dout = rng.normal(size=(n, out_dim)).astype(np.float32)
    ks_np = [k for k, _ in layers] + [out[0][0]]
    bs_np = [b for _, b in layers] + [out[0][1]]
    blob = ops.pack_mlp128_train_weights(ks_np, bs_np, nfx_lib.IN_XYZ, out_dim).to(cuda)
    dks = [torch.zeros(k.shape, device=cuda) for k in ks_np]
    dbs = [torch.zeros(b.shape, device=cuda) for b in bs_np]
    ops.mlp128_bwd(nfx_lib.IN_XYZ, dev(xyz, cuda), dev(dout, cuda), blob, dks, dbs, out_act=act,
                   xyz_scale=0.9, post_scale=scale)
    # (1) against autograd through the SAME bf16-rounded forward (pins the kernel's logic: ReLU masks of
    #     a low-precision forward differ from the fp64 ones for units near zero, which alone moves the
    #     gradient by several % in Frobenius norm), (2) loosely against the plain fp64 network.
    for quant, tol in ((True, 2.5e-2), (False, 0.2)):
        ks, bs = _params(layers, out, 'cpu')
        pe = torch_embed((torch.tensor(xyz) * np.float32(0.9)).double(), 10)
        y = scale * torch_mlp128(pe, ks, bs, act, quant)
        y.backward(torch.tensor(dout, dtype=torch.float64))
        _check_grads(dks, [k.grad for k in ks], 'dkernel', tol)
        _check_grads(dbs, [b.grad for b in bs], 'dbias', tol)
    # accumulation semantics: a second call doubles the gradients
    ops.mlp128_bwd(nfx_lib.IN_XYZ, dev(xyz, cuda), dev(dout, cuda), blob, dks, dbs, out_act=act,
                   xyz_scale=0.9, post_scale=scale)
    _check_grads([d / 2 for d in dks], [k.grad for k in ks], 'dkernel x2', 0.2)


def test_lvis_backward_vs_autograd(nfx_lib, cuda):
    from nerfactor_amd import ops
    layers, out = net128(90, 90, 1)
    n = 21
    rng, lxyz, _, xyz, _, _ = scene(n, 91)
    xyz_j = xyz + rng.normal(size=xyz.shape).astype(np.float32) * 0.01
    dout = rng.normal(size=(n, 512)).astype(np.float32)
    ks_np = [k for k, _ in layers] + [out[0][0]]
    bs_np = [b for _, b in layers] + [out[0][1]]
    blob = ops.pack_mlp128_train_weights(ks_np, bs_np, nfx_lib.IN_XYZ_LDIR, 1).to(cuda)
    dks = [torch.zeros(k.shape, device=cuda) for k in ks_np]
    dbs = [torch.zeros(b.shape, device=cuda) for b in bs_np]
    # the jittered call: MLP evaluated at xyz_j, directions taken from xyz (nerfactor.py:195,226)
    ops.mlp128_bwd(nfx_lib.IN_XYZ_LDIR, dev(xyz_j, cuda), dev(dout, cuda), blob, dks, dbs, out_act='sigmoid',
                   lxyz=dev(lxyz, cuda), xyz_dir=dev(xyz, cuda))
    surf2l = torch.tensor(R.calc_ldir(xyz, lxyz), dtype=torch.float64).reshape(-1, 3)
    pts = torch.tensor(xyz_j, dtype=torch.float64)[:, None, :].expand(n, 512, 3).reshape(-1, 3)
    x = torch.cat((torch_embed(pts, 10), torch_embed(surf2l, 4)), -1)
    for quant, tol in ((True, 2.5e-2), (False, 0.2)):
        ks, bs = _params(layers, out, 'cpu')
        y = torch_mlp128(x, ks, bs, 'sigmoid', quant).reshape(n, 512)
        y.backward(torch.tensor(dout, dtype=torch.float64))
        _check_grads(dks, [k.grad for k in ks], 'dkernel', tol)
        _check_grads(dbs, [b.grad for b in bs], 'dbias', tol)


def test_amsgrad_matches_keras_semantics(nfx_lib, cuda):
    from nerfactor_amd import ops
    rng = np.random.default_rng(7)
    n = 10007
    p = rng.normal(size=n).astype(np.float32)
    m = np.zeros(n, np.float32)
    v = np.zeros(n, np.float32)
    vh = np.zeros(n, np.float32)
    tp, tm, tv, tvh = (dev(a, cuda) for a in (p, m, v, vh))
    lr, b1, b2, eps = 5e-3, 0.9, 0.999, 1e-7
    for step in range(1, 6):
        g = (rng.normal(size=n) * (10. if step == 2 else 1.)).astype(np.float32)
        ops.amsgrad_step(tp, dev(g, cuda), tm, tv, tvh, lr, step)
        lr_t = lr * np.sqrt(1 - b2 ** step) / (1 - b1 ** step)
        f = np.float32  # TF evaluates the update in float32, including (1 - beta)
        m = f(b1) * m + (f(1) - f(b1)) * g
        v = f(b2) * v + (f(1) - f(b2)) * (g * g)
        vh = np.maximum(vh, v)
        p = p - f(lr_t) * m / (np.sqrt(vh) + f(eps))
    np.testing.assert_allclose(tp.cpu().numpy(), p, rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(tvh.cpu().numpy(), vh, rtol=1e-5)
