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


def test_transposing_lds_read_contracts_over_rows(nfx_lib, cuda):
    """csrc/tr16.hpp: two row-major [16 rows][32 slots] tiles -> LDS -> ds_read_b64_tr_b16 -> one MFMA whose K axis is
    the ROW axis must give H^T Z (random, non-symmetric tiles: a transposed or permuted operand cannot pass), and the
    instruction's raw lane map is the one the header documents."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(5)
    h = rng.integers(-8, 9, size=(16, 32)).astype(np.float32)      # exact in bf16, products exact in fp32
    z = rng.integers(-8, 9, size=(16, 32)).astype(np.float32)
    got = ops.selftest_tr16(dev(h, cuda), dev(z, cuda)).cpu().numpy()
    np.testing.assert_array_equal(got, h.T @ z)
    raw = ops.selftest_tr16().cpu().numpy().astype(np.int64)           # lane l read elements 4 l .. 4 l + 3 of its block
    want = np.zeros((64, 4), np.int64)
    for l in range(64):
        i, base = l & 15, 64 * (l >> 4)                                # 16-lane group = a [4 rows][16 slots] block
        for j in range(4):
            want[l, j] = base + 16 * j + i                             # slot i of row j
    np.testing.assert_array_equal(raw, want)


@pytest.mark.parametrize("fused", [1, 0])
@pytest.mark.parametrize("out_dim,act,scale,n", [(3, 'sigmoid', .77, 1000), (3, None, 1., 77), (1, 'sigmoid', 1., 300)])
def test_mlp128_xyz_backward_vs_autograd(nfx_lib, cuda, nfx_opt, out_dim, act, scale, n, fused):
    from nerfactor_amd import ops
    nfx_opt.set("wgrad_fused", fused)     # 1: weight gradients accumulated on chip (mlp128_bwd_fused.hip); 0: the r03 path
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


@pytest.mark.parametrize("path,n", [("fused", 21), ("fused3", 23), ("gemm0", 21), ("gemm1", 21)])
def test_lvis_backward_vs_autograd(nfx_lib, cuda, nfx_opt, path, n):
    from nerfactor_amd import ops
    if path.startswith("fused"):          # weight gradients accumulated on chip, two launches + one ordered reduction
        nfx_opt.set("wgrad_fused", 1)
        if path == "fused3":              # 3 persistent workgroups: 30 tiles each, the weight ring wraps, a tail tile
            nfx_opt.set("m128_blocks", 3)
    else:                                 # the r03 path: activations stored, both weight-gradient GEMM kernels (train.hip)
        nfx_opt.set("wgrad_fused", 0)
        nfx_opt.set("wgrad_lds", path[-1])
    layers, out = net128(90, 90, 1)
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


def test_heads_of_one_launch_equal_one_launch_per_head(nfx_lib, cuda, nfx_opt):
    """nfx_mlp128_bwd_heads (round 5: the three xyz heads of a NeRFactor step in ONE launch pair + reduction, blockIdx.y =
    head) against nfx_mlp128_bwd per head: the same bits in every gradient tensor, for a persistent grid smaller than the
    tile count too; gradient buffers shared between heads are refused; autograd's deferred form (autograd.Mlp128Xyz records
    the heads, an engine callback launches them together) gives the same parameter gradients as the immediate one."""
    from nerfactor_amd import autograd, ops
    rng, _, _, xyz, _, _ = scene(700, 96)
    outs = [(3, None, 1.0), (3, 'sigmoid', 0.7), (1, 'softplus', 1.0), (3, 'sigmoid', 1.0)]
    nets = []
    for seed, (od, act, scale) in enumerate(outs):
        layers, out = net128(40 + seed, 63, od)
        ks = [k for k, _ in layers] + [out[0][0]]
        bs = [b for _, b in layers] + [out[0][1]]
        nets.append((ks, bs, ops.pack_mlp128_train_weights(ks, bs, nfx_lib.IN_XYZ, od).to(cuda),
                     dev(rng.normal(size=(700, od)), cuda), act, scale))
    x = dev(xyz, cuda)

    def zeros(ks, bs):
        return [torch.zeros(k.shape, device=cuda) for k in ks], [torch.zeros(b.shape, device=cuda) for b in bs]
    for blocks in (None, 3):
        if blocks:
            nfx_opt.set("m128_blocks", blocks)
        else:
            nfx_opt.unset("m128_blocks")
        want, heads, got = [], [], []
        for ks, bs, blob, dout, act, scale in nets:
            dks, dbs = zeros(ks, bs)
            ops.mlp128_bwd(nfx_lib.IN_XYZ, x, dout, blob, dks, dbs, out_act=act, xyz_scale=0.9, post_scale=scale)
            want.append(dks + dbs)
            dks, dbs = zeros(ks, bs)
            heads.append((dout, blob, dks, dbs, act, scale))
            got.append(dks + dbs)
        ops.mlp128_bwd_heads(nfx_lib.IN_XYZ, x, heads, xyz_scale=0.9)
        for h, (g, w) in enumerate(zip(got, want)):
            assert all(torch.equal(a, b) for a, b in zip(g, w)) and float(g[0].abs().sum()) > 0, (blocks, h)
    nfx_opt.unset("m128_blocks")
    with pytest.raises(Exception, match='share a gradient buffer'):
        ops.mlp128_bwd_heads(nfx_lib.IN_XYZ, x, [heads[0], heads[0]], xyz_scale=0.9)
    with pytest.raises(Exception, match='heads'):
        ops.mlp128_bwd_heads(nfx_lib.IN_XYZ, x, heads + heads[:1], xyz_scale=0.9)

    # autograd: parameters that own their gradient buffers (as under optim.AMSGrad) -> deferred, batched; else immediate
    def grads(batch):
        autograd.BATCH_HEADS = batch
        try:
            params, loss = [], 0.
            for ks, bs, blob, dout, act, scale in nets[:3]:
                P = [torch.nn.Parameter(dev(k, cuda)) for k in ks] + [torch.nn.Parameter(dev(b, cuda)) for b in bs]
                for q in P:
                    q.grad = torch.zeros_like(q)
                fwd = ops.pack_mlp128_weights(ks, bs, nfx_lib.IN_XYZ, dout.shape[1]).to(cuda)
                y = autograd.Mlp128Xyz.apply(x, fwd, lambda blob=blob: blob, 'bf16', dout.shape[1], act, 0.9, scale, 0., *P)
                loss = loss + (y * dout).sum()
                params.append(P)
            loss.backward()
            assert not autograd._heads['pending'] and not autograd._heads['armed']
            return [q.grad.clone() for P in params for q in P]
        finally:
            autograd.BATCH_HEADS = True
    a, b = grads(True), grads(False)
    assert all(torch.equal(u, v) for u, v in zip(a, b)) and float(a[0].abs().sum()) > 0


@pytest.mark.parametrize("in_kind,n", [("xyz", 1500), ("lvis", 37)])
def test_fused_weight_gradients_match_the_gemm_path(nfx_lib, cuda, nfx_opt, in_kind, n):
    """Same bf16 products, fp32 sums in a different order: the fused kernels (slot-permuted accumulators, two launches,
    workgroup-ordered reduction) against the stored-activation path, every gradient tensor within 2e-4 relative
    Frobenius — an operand permutation or a missing block would be off by O(1) — and twice the same bits."""
    from nerfactor_amd import ops
    lv = in_kind == "lvis"
    layers, out = net128(95, 90 if lv else 63, 1 if lv else 3)
    rng, lxyz, _, xyz, _, _ = scene(n, 96)
    xyz_j = xyz + rng.normal(size=xyz.shape).astype(np.float32) * 0.01
    dout = rng.normal(size=(n, 512 if lv else 3)).astype(np.float32)
    ks_np = [k for k, _ in layers] + [out[0][0]]
    bs_np = [b for _, b in layers] + [out[0][1]]
    kind = nfx_lib.IN_XYZ_LDIR if lv else nfx_lib.IN_XYZ
    blob = ops.pack_mlp128_train_weights(ks_np, bs_np, kind, 1 if lv else 3).to(cuda)

    def run(fused, blocks=None):
        nfx_opt.set("wgrad_fused", fused)
        if blocks:
            nfx_opt.set("m128_blocks", blocks)
        else:
            nfx_opt.unset("m128_blocks")
        dks = [torch.zeros(k.shape, device=cuda) for k in ks_np]
        dbs = [torch.zeros(b.shape, device=cuda) for b in bs_np]
        kw = dict(lxyz=dev(lxyz, cuda), xyz_dir=dev(xyz, cuda)) if lv else {}
        ops.mlp128_bwd(kind, dev(xyz_j, cuda), dev(dout, cuda), blob, dks, dbs, out_act='sigmoid', xyz_scale=0.9, **kw)
        return dks + dbs
    want = run(0)
    for blocks in (None, 5):
        got = run(1, blocks)
        for i, (g, w) in enumerate(zip(got, want)):
            rel = float((g - w).norm() / (w.norm() + 1e-20))
            assert rel < 2e-4, (in_kind, blocks, 'tensor %d' % i, rel)
        again = run(1, blocks)
        assert all(torch.equal(a, b) for a, b in zip(got, again)), "fused weight gradients are not bit-reproducible"


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


def _torch_render(xyz, cam, normal, albedo, rough, lvis, lxyz, lareas, light, to_srgb):
    """Differentiable float64 restatement of nerfactor.py:315-342 + microfacet.py (torch autograd)."""
    from nerfactor_amd.brdf.microfacet.microfacet import Microfacet
    from nerfactor_amd.nerfactor.util import img as imgutil

    def nz(x, dim):
        return x * torch.rsqrt(torch.clamp((x * x).sum(dim, keepdim=True), min=1e-6))
    surf2l = nz(lxyz[None] - xyz[:, None], 2)
    surf2c = nz(cam - xyz, 1)
    brdf = Microfacet(f0=0.04)(surf2l, surf2c, normal, albedo=albedo, rough=rough)
    cos = torch.einsum('ijk,ik->ij', surf2l, normal)
    lv = (cos > 0).double() * lvis
    rgb = (brdf * (lv[:, :, None] * light[None]) * cos[:, :, None] * lareas[None, :, None]).sum(1)
    rgb = torch.clamp(rgb, 0., 1.)
    return imgutil.linear2srgb(rgb) if to_srgb else rgb


@pytest.mark.parametrize("to_srgb", [True, False])
def test_shade_backward_vs_autograd(nfx_lib, cuda, to_srgb):
    from nerfactor_amd import ops
    from tests.test_gpu_nerfactor import _shade_inputs
    n = 64
    rng, lxyz, lareas, xyz, cam, normal, albedo, rough, lvis, lights = _shade_inputs(n, 95)
    rough = np.clip(rough, 0.25, 1.)  # keep GGX away from its fp32-ill-conditioned corner
    light = (lights[0].reshape(512, 3) * 0.5).astype(np.float32)
    drgb = rng.normal(size=(n, 3)).astype(np.float32)
    d_light = torch.zeros(512, 3, device=cuda)
    got = ops.shade_bwd(dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(albedo, cuda), dev(lvis, cuda),
                        dev(lxyz, cuda), dev(lareas, cuda), dev(light, cuda), dev(drgb, cuda), d_light,
                        rough=dev(rough, cuda), f0=0.04, linear2srgb=to_srgb)
    t = lambda a, g=False: torch.tensor(a, dtype=torch.float64, requires_grad=g)
    tn, ta, tr, tv, tl = t(normal, True), t(albedo, True), t(rough, True), t(lvis, True), t(light, True)
    rgb = _torch_render(t(xyz), t(cam), tn, ta, tr, tv, t(lxyz), t(lareas.reshape(-1)), tl, to_srgb)
    rgb.backward(t(drgb))
    inside = ((rgb.detach() > 1e-4) & (rgb.detach() < 1 - 1e-4)).all(1).numpy()  # away from the clip kinks
    assert inside.mean() > 0.5

    def rel(g, w):
        g, w = g.double().cpu().numpy()[inside], w.numpy()[inside]
        return np.linalg.norm(g - w) / (np.linalg.norm(w) + 1e-30)
    assert rel(got[0], ta.grad) < 1e-3, rel(got[0], ta.grad)             # d albedo
    assert rel(got[2], tv.grad) < 1e-3, rel(got[2], tv.grad)             # d lvis
    assert rel(got[3][:, None], tr.grad) < 2e-2, rel(got[3][:, None], tr.grad)   # d rough
    assert rel(got[1], tn.grad) < 2e-2, rel(got[1], tn.grad)             # d normal
    # d light: sum over the points that are inside the clip range only -> recompute with those points
    d_light2 = torch.zeros(512, 3, device=cuda)
    sel = torch.from_numpy(np.nonzero(inside)[0]).to(cuda)
    pick = lambda a: dev(a, cuda)[sel].contiguous()
    ops.shade_bwd(pick(xyz), pick(cam), pick(normal), pick(albedo), pick(lvis), dev(lxyz, cuda), dev(lareas, cuda),
                  dev(light, cuda), pick(drgb), d_light2, rough=pick(rough), f0=0.04, linear2srgb=to_srgb)
    tl2 = t(light, True)
    idx = np.nonzero(inside)[0]
    rgb2 = _torch_render(t(xyz[idx]), t(cam[idx]), t(normal[idx]), t(albedo[idx]), t(rough[idx]), t(lvis[idx]),
                         t(lxyz), t(lareas.reshape(-1)), tl2, to_srgb)
    rgb2.backward(t(drgb[idx]))
    e = np.linalg.norm(d_light2.double().cpu().numpy() - tl2.grad.numpy()) / np.linalg.norm(tl2.grad.numpy())
    assert e < 1e-3, e


# ------------------------------------------------------------------------ full training step
def _torch_reference_grads(model, np_batch, xyz_noise, global_bs):
    """float64 torch autograd of the nerfactor_microfacet training loss with the model's weights:
    the reference semantics of nerfactor.py:181-541 restated with differentiable torch ops."""
    cam, rgb, alpha, xyz, normal, lvis = [torch.tensor(a, dtype=torch.float64) for a in np_batch]
    cfg = model.config
    mask = alpha[:, 0] > 0
    P = {n: p.detach().double().cpu().requires_grad_(True) for n, p in model.named_parameters() if p.requires_grad}

    def net(name, x, act):
        ks = [P['net_%s_mlp_layer%d.kernel' % (name, i)] for i in range(4)] + [P['net_%s_out_layer0.kernel' % name]]
        bs = [P['net_%s_mlp_layer%d.bias' % (name, i)] for i in range(4)] + [P['net_%s_out_layer0.bias' % name]]
        return torch_mlp128(x, ks, bs, act)

    lxyz = model.lxyz.double().cpu().reshape(-1, 3)
    lareas = model.lareas.double().cpu().reshape(-1)
    xm, cm = xyz[mask], cam[mask]
    surf2l = R_t_normalize(lxyz[None] - xm[:, None], 2)

    def heads(p):
        pe = torch_embed(p, 10)
        nrm = R_t_normalize(net('normal', pe, None) + 1e-6, 1)
        x = torch.cat((torch_embed(p[:, None, :].expand(-1, 512, -1).reshape(-1, 3), 10),
                       torch_embed(surf2l.reshape(-1, 3), 4)), -1)
        lv = net('lvis', x, 'sigmoid').reshape(-1, 512)
        alb = 0.77 * net('albedo', pe, 'sigmoid') + 0.03
        z = net('brdf_z', pe, 'sigmoid')
        return nrm, lv, alb, z

    nrm, lv, alb, z = heads(xm)
    nrm_j, lv_j, alb_j, z_j = heads(xm + torch.tensor(xyz_noise, dtype=torch.float64))
    light = torch.clamp(P['_light'], min=0.)
    rgb_pred = _torch_render(xm, cm, nrm, alb, z, lv, lxyz, lareas, light.reshape(-1, 3), True)
    n_all = alpha.shape[0]

    def full(v):
        out = torch.zeros((n_all,) + tuple(v.shape[1:]), dtype=torch.float64)
        out[mask] = v
        return out

    def on_bg(x):
        return x * alpha + 1. * (1. - alpha)
    mse = lambda a, b: ((a - b) ** 2).mean(-1)
    mae = lambda a, b: (a - b).abs().mean(-1)
    rgb_p, rgb_g = on_bg(full(rgb_pred)), on_bg(full(rgb[mask]))
    n_p, n_g = on_bg(full(nrm)), on_bg(full(normal[mask]))
    v_p, v_g = on_bg(full(lv)), on_bg(full(lvis[mask]))
    loss = mse(rgb_g, rgb_p) + 0.1 * mse(n_g, n_p) + 0.1 * mse(v_g, v_p)
    loss = loss + 0.05 * mae(n_p, full(nrm_j)) + 0.05 * mae(v_p, full(lv_j))
    loss = loss + 0.05 * mae(full(alb), full(alb_j)) + 0. * mae(full(z), full(z_j))
    dx = light - torch.roll(light, 1, 1)
    dy = light - torch.roll(light, 1, 0)
    loss = loss + 5e-6 * (dx ** 2 + dy ** 2).sum()
    weighted = loss.sum() / global_bs
    weighted.backward()
    return float(weighted), {n: p.grad for n, p in P.items()}


def R_t_normalize(x, dim):
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim, keepdim=True), min=1e-6))


def test_nerfactor_microfacet_train_step_vs_autograd(nfx_lib, cuda):
    """One full training step through the plugin (forward kernels -> loss -> backward kernels -> one
    bucket -> fused AMSGrad) vs float64 torch autograd of the same loss: gradient direction / norm per
    parameter tensor, the loss value, and the parameters after the update."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from tests.test_gpu_nerfactor import _nerfactor_batch
    cfg = make_config('nerfactor_microfacet', shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='')
    torch.manual_seed(3)
    model = get_model_class('nerfactor_microfacet')(cfg).to(cuda)
    with torch.no_grad():  # non-zero biases, moderate roughness
        for n_, p in model.named_parameters():
            if n_.endswith('bias'):
                p.uniform_(-0.2, 0.2)
    n = 160
    np_batch, t_batch, lxyz, lareas = _nerfactor_batch(n, 97, cuda)
    n_fg = int((np_batch[2][:, 0] > 0).sum())
    noise = np.random.default_rng(98).normal(size=(n_fg, 3)).astype(np.float32) * 0.01
    global_bs = n
    want_loss, want = _torch_reference_grads(model, np_batch, noise, global_bs)
    opt = optim.make_optimizer(model, cfg)
    before = opt.flat.clone()
    opt.zero_grad()
    pred, gt, loss_kwargs, _ = model(t_batch, mode='train', xyz_noise=dev(noise, cuda))
    loss_kwargs['keep_batch'] = True
    weighted = model.compute_loss(pred, gt, **loss_kwargs).sum() / global_bs
    weighted.backward()
    assert abs(float(weighted) - want_loss) < 2e-2 * abs(want_loss)
    report = {}
    for name, p in model.named_parameters():
        if not p.requires_grad:
            continue
        g, w = p.grad.double().cpu().reshape(-1), want[name].reshape(-1)
        cos = float((g @ w) / (g.norm() * w.norm() + 1e-30))
        report[name] = (round(cos, 4), round(float(g.norm() / (w.norm() + 1e-30)), 3))
    bad = {k: v for k, v in report.items() if v[0] < 0.97 or not 0.85 < v[1] < 1.15}
    assert not bad, bad
    total = opt.step(loss=weighted.detach())
    assert abs(float(total) - float(weighted)) < 1e-6
    moved = (opt.flat - before).abs()
    # first Adam step: lr_t * m / sqrt(vhat) = lr * sign(g) wherever |g| >> eps
    assert 0.9 * 5e-3 < float(moved.max()) <= 5e-3 * 1.001
    assert opt.iterations == 1


# --------------------------------------------------------------- learned-BRDF backward (frozen prior)
class _SafeAcos(torch.autograd.Function):  # util/math.py:41-60
    @staticmethod
    def forward(ctx, x):
        xc = torch.clamp(x, -1., 1.)
        ctx.save_for_backward(xc)
        return torch.acos(xc)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        return dy * (-1. / (torch.sqrt(1. - xc ** 2 + 1e-6) + 1e-6))


class _SafeAtan2(torch.autograd.Function):  # util/math.py:24-38
    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return torch.atan2(x, y)

    @staticmethod
    def backward(ctx, dz):
        x, y = ctx.saved_tensors
        den = x ** 2 + y ** 2 + 1e-6
        return dz * y / den, dz * (-x / den)


def _torch_learned_spec(normal, xyz, cam, z, lxyz, ks, bs, quant):
    """Differentiable float64 restatement of nerfactor.py:413-458 + util/geom.py:119-192."""
    nz = R_t_normalize
    n_pts, nl = xyz.shape[0], lxyz.shape[0]
    surf2l = nz(lxyz[None] - xyz[:, None], 2)
    surf2c = nz(cam - xyz, 1)
    nn = nz(normal, 1)
    zc = torch.tensor([1e-6, 1e-6, 1 + 1e-6], dtype=torch.float64).expand_as(nn)
    t = nz(torch.cross(nn, zc, dim=1), 1)
    b = nz(torch.cross(nn, t, dim=1), 1)
    rot = torch.stack((t, b, nn), 1)
    vdir = torch.einsum('jkl,jl->jk', rot, surf2c)
    ldir = torch.einsum('jkl,jnl->jnk', rot, surf2l).reshape(-1, 3)
    vrep = vdir[:, None, :].expand(-1, nl, -1).reshape(-1, 3)
    a, bb = nz(ldir, 1), nz(vrep, 1)
    h = nz((a + bb) / 2, 1)
    theta_h = _SafeAcos.apply(h[:, 2])
    phi_h = _SafeAtan2.apply(h[:, 1], h[:, 0])

    def rot_vec(v, axis, ang):
        axis = torch.tensor(axis, dtype=torch.float64).reshape(1, 3)
        c, s = torch.cos(ang)[:, None], torch.sin(ang)[:, None]
        return v * c + axis * (v @ axis.T) * (1 - c) + torch.cross(axis.expand_as(v), v, dim=1) * s
    diff = rot_vec(rot_vec(bb, (0., 0., 1.), -phi_h), (0., 1., 0.), -theta_h)
    theta_d = _SafeAcos.apply(diff[:, 2])
    phi_d = torch.remainder(_SafeAtan2.apply(diff[:, 1], diff[:, 0]), np.pi)
    rus = torch.stack((phi_d, theta_h, theta_d), 1)
    zrep = z[:, None, :].expand(-1, nl, -1).reshape(-1, z.shape[1])
    x = torch.cat((zrep, torch_embed(rus, 2)), 1)
    y = torch.nn.functional.softplus(torch_mlp128(x, ks, bs, None, quant))[:, 0]
    front = (ldir[:, 2] > 0).double()
    return (y * front).reshape(n_pts, nl), ldir[:, 2].reshape(n_pts, nl)


@pytest.mark.parametrize("zd", [3, 1])
def test_brdf_spec_backward_vs_autograd(nfx_lib, cuda, zd):
    from nerfactor_amd import ops
    layers, out = net128(110 + zd, zd + 15, 1)
    ks_np = [k for k, _ in layers] + [out[0][0]]
    bs_np = [b for _, b in layers] + [out[0][1]]
    blob = ops.pack_brdf_train_weights(ks_np, bs_np, zd).to(cuda)
    n = 40
    rng, lxyz, _, xyz, cam, normal = scene(n, 111)
    zl = rng.normal(size=(n, zd)).astype(np.float32)
    dspec = rng.normal(size=(n, 512)).astype(np.float32)
    d_z, d_n = ops.brdf_spec_bwd(dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(zl, cuda), dev(lxyz, cuda),
                                 blob, dev(dspec, cuda))
    t = lambda a, g=False: torch.tensor(a, dtype=torch.float64, requires_grad=g)
    for quant, tol in ((True, 3e-2), (False, 0.25)):
        ks = [t(k) for k in ks_np]
        bs = [t(b) for b in bs_np]
        tn, tz = t(normal, True), t(zl, True)
        spec, lz = _torch_learned_spec(tn, t(xyz), t(cam), tz, t(lxyz), ks, bs, quant)
        # rows whose front-lit test sits on the fp32 knife edge are excluded on both sides
        stable = (lz.detach().abs() > 1e-4).double()
        (spec * t(dspec) * stable).sum().backward()
        # (the kernel sums every row: the unstable ones contribute O(1e-4 * n) here)
        for name, got, want in (('d_z', d_z, tz.grad), ('d_normal', d_n, tn.grad)):
            g, w = got.double().cpu().numpy(), want.numpy()
            err = np.linalg.norm(g - w) / (np.linalg.norm(w) + 1e-30)
            assert err < tol, (name, quant, err)


@pytest.mark.determinism
@pytest.mark.parametrize("zd,n,zero_frac", [(3, 700, 0.5), (1, 33, 0.5), (3, 257, 0.97), (3, 64, 0.0)])
def test_brdf_spec_backward_over_the_rows_with_a_gradient_equals_every_row(nfx_lib, cuda, nfx_opt, zd, n, zero_frac):
    """nfx_brdf_spec_bwd_rows (round 6): the learned BRDF differentiated over the (point, light) rows with d spec != 0 only
    — the shading backward zeroes the back-facing half — against the same call over EVERY row (option brdf_bwd_rows = 0):
    both sum in fixed point from the first addition on, so d z and d normal are the same bits whatever the grouping of rows
    into waves and whatever order the row list came out in (run three times).  All-zero and no-zero gradients included."""
    from nerfactor_amd import ops
    layers, out = net128(110 + zd, zd + 15, 1)
    blob = ops.pack_brdf_train_weights([k for k, _ in layers] + [out[0][0]], [b for _, b in layers] + [out[0][1]], zd).to(cuda)
    rng, lxyz, _, xyz, cam, normal = scene(n, 211)
    zl = rng.normal(size=(n, zd)).astype(np.float32)
    dspec = rng.normal(size=(n, 512)).astype(np.float32)
    dspec[rng.uniform(size=dspec.shape) < zero_frac] = 0.
    if n == 33:
        dspec[5] = 0.           # a whole point without gradient
    args = (dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(zl, cuda), dev(lxyz, cuda), blob, dev(dspec, cuda))
    nfx_opt.set("brdf_bwd_rows", "0")
    dz_all, dn_all = ops.brdf_spec_bwd(*args)
    nfx_opt.set("brdf_bwd_rows", "1")
    for _ in range(3):
        dz, dn = ops.brdf_spec_bwd(*args)
        assert torch.equal(dz, dz_all) and torch.equal(dn, dn_all)
    assert bool(torch.isfinite(dz_all).all()) and (zero_frac == 0.97 or float(dn_all.abs().max()) > 0)
    if n == 33:
        assert not bool(dz_all[5].any()) and not bool(dn_all[5].any())
    zero = torch.zeros_like(args[-1])
    dz0, dn0 = ops.brdf_spec_bwd(*args[:-1], zero)
    assert not bool(dz0.any()) and not bool(dn0.any())


def test_nerfactor_learned_brdf_train_step_runs_and_descends(nfx_lib, cuda):
    """The flagship model (frozen learned BRDF) trains end to end through libnfx: a few steps on a fixed
    batch must decrease the loss, and every trainable tensor must receive a finite, non-zero gradient."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from tests.test_gpu_nerfactor import _nerfactor_batch
    cfg = make_config('nerfactor', shape_mode='finetune', shape_model_ckpt='none', brdf_model_ckpt='none',
                      test_envmap_dir='', lr='2e-3')
    torch.manual_seed(4)
    model = get_model_class('nerfactor')(cfg).to(cuda)
    opt = optim.make_optimizer(model, cfg)
    _, t_batch, _, _ = _nerfactor_batch(256, 120, cuda)
    torch.manual_seed(0)
    losses = []
    for step in range(8):
        torch.manual_seed(1)  # same jitter every step: a deterministic objective
        loss, _ = optim.train_step(model, t_batch, opt, 256)
        losses.append(float(loss))
        if step == 0:
            for name, p in model.named_parameters():
                if p.requires_grad:
                    assert torch.isfinite(p.grad).all() and float(p.grad.abs().max()) > 0, name
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
    assert not any(p.requires_grad for p in model.brdf_model.parameters())


# ---------------------------------------------------------------------------------------- NeRF training path
def torch_nerf(pe_x, pe_v, ks, bs, quant):
    """rgbs[M,4] of nerf.py:256-290 (enc 8x256 with the skip after layer 4, sigma_out, bottleneck, rgb_out)."""
    q = q16 if quant else (lambda t: t)
    h = pe_x
    for i in range(8):
        h = torch.relu(q(h) @ q(ks[i]) + bs[i])
        if i == 4:
            h = torch.cat((h, pe_x), -1)
    sigma = q(h) @ q(ks[8]) + bs[8]
    bott = q(h) @ q(ks[9]) + bs[9]
    r = torch.relu(q(torch.cat((bott, pe_v), -1)) @ q(ks[10]) + bs[10])
    return torch.cat((q(r) @ q(ks[11]) + bs[11], sigma), -1)


def torch_composite(rgbs, z, rayd, white_bg, noise=None):
    """nerf.py:184-254 in torch (any dtype), for autograd."""
    dist = torch.cat((z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)), 1) * rayd.norm(dim=1, keepdim=True)
    sg = rgbs[..., 3] if noise is None else rgbs[..., 3] + noise
    alpha = 1. - torch.exp(-torch.relu(sg) * dist)
    t = 1. - alpha + 1e-6
    T = torch.cat((torch.ones_like(t[:, :1]), torch.cumprod(t, 1)[:, :-1]), 1)
    w = alpha * T
    occu = w.sum(1, keepdim=True)
    rgb = (w[..., None] * torch.sigmoid(rgbs[..., :3])).sum(1)
    return rgb * occu + (1. if white_bg else 0.) * (1. - occu)


@pytest.mark.parametrize("n,s,white_bg,use_noise", [(37, 64, True, False), (9, 192, False, True), (5, 7, True, False)])
def test_composite_backward_vs_autograd(nfx_lib, cuda, n, s, white_bg, use_noise):
    from nerfactor_amd import ops
    rng = np.random.default_rng(300 + s)
    rgbs = rng.normal(size=(n, s, 4)).astype(np.float32)
    rgbs[..., 3] = rgbs[..., 3] * 3. + 0.5
    z = np.sort(rng.uniform(2., 6., size=(n, s)).astype(np.float32), 1)
    rayd = rng.normal(size=(n, 3)).astype(np.float32)
    noise = rng.normal(size=(n, s)).astype(np.float32) if use_noise else None
    g = rng.normal(size=(n, 3)).astype(np.float32)
    got = ops.composite_bwd(dev(rgbs, cuda), dev(z, cuda), dev(rayd, cuda), dev(g, cuda), white_bg=white_bg,
                            noise=None if noise is None else dev(noise, cuda)).cpu().numpy()
    t = torch.tensor(rgbs, dtype=torch.float64, requires_grad=True)
    out = torch_composite(t, torch.tensor(z, dtype=torch.float64), torch.tensor(rayd, dtype=torch.float64), white_bg,
                          None if noise is None else torch.tensor(noise, dtype=torch.float64))
    out.backward(torch.tensor(g, dtype=torch.float64))
    want = t.grad.numpy()
    assert np.isfinite(got).all()
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-5 * np.abs(want).max())
    # and the forward the backward re-computes is the forward kernel's
    rgb = ops.composite_fwd(dev(rgbs, cuda), dev(z, cuda), dev(rayd, cuda), white_bg=white_bg,
                            noise=None if noise is None else dev(noise, cuda))[0].cpu().numpy()
    np.testing.assert_allclose(rgb, out.detach().numpy(), atol=2e-5)


@pytest.mark.parametrize("wgrad_lds", ["0", "1"])
@pytest.mark.parametrize("n_rays,s", [(40, 7), (3, 192)])
def test_nerf_mlp_backward_vs_autograd(nfx_lib, cuda, n_rays, s, nfx_opt, wgrad_lds):
    from nerfactor_amd import ops
    nfx_opt.set("wgrad_lds", wgrad_lds)
    from tests import common
    net = common.nerf_nets(seed=5, opaque=False)[0]
    ks_np, bs_np = common.nerf_layers(net)
    rng = np.random.default_rng(310 + s)
    rayo = rng.uniform(-1, 1, size=(n_rays, 3)).astype(np.float32)
    rayd = rng.normal(size=(n_rays, 3)).astype(np.float32)
    rayd /= np.linalg.norm(rayd, axis=1, keepdims=True)
    z = np.sort(rng.uniform(0.5, 3., size=(n_rays, s)).astype(np.float32), 1)
    d_rgbs = rng.normal(size=(n_rays, s, 4)).astype(np.float32)
    blob = ops.pack_nerf_train_weights(ks_np, bs_np).to(cuda)
    dks = [torch.zeros(k.shape, device=cuda) for k in ks_np]
    dbs = [torch.zeros(b.shape, device=cuda) for b in bs_np]
    ops.nerf_mlp_bwd(dev(rayo, cuda), dev(rayd, cuda), dev(z, cuda), dev(d_rgbs, cuda), blob, dks, dbs)
    pts = (rayo[:, None, :] + rayd[:, None, :] * z[:, :, None]).reshape(-1, 3)      # fp32, as the kernel forms them
    views = np.broadcast_to(rayd[:, None, :], (n_rays, s, 3)).reshape(-1, 3)
    pe_x = torch_embed(torch.tensor(pts).double(), 10)
    pe_v = torch_embed(torch.tensor(views).double(), 4)
    for quant, tol in ((True, 3e-2), (False, 0.3)):
        ks = [torch.tensor(k, dtype=torch.float64, requires_grad=True) for k in ks_np]
        bs = [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in bs_np]
        y = torch_nerf(pe_x, pe_v, ks, bs, quant)
        y.backward(torch.tensor(d_rgbs.reshape(-1, 4), dtype=torch.float64))
        _check_grads(dks, [k.grad for k in ks], 'dkernel(quant=%s)' % quant, tol)
        _check_grads(dbs, [b.grad for b in bs], 'dbias(quant=%s)' % quant, tol)
    # the train blob's forward half is the inference blob
    inf = ops.pack_nerf_weights(ks_np, bs_np)
    nfrag = 1272 * 1024
    assert torch.equal(inf[:nfrag], blob[:nfrag].cpu()) and torch.equal(inf[nfrag:], blob[-(inf.numel() - nfrag):].cpu())


@pytest.mark.parametrize("n_rays,s,keep", [(150, 192, 0.4), (700, 64, 0.03), (129, 130, 1.0), (90, 192, 0.)])
def test_nerf_mlp_backward_over_the_points_with_a_gradient(nfx_lib, cuda, nfx_opt, n_rays, s, keep):
    """nfx_nerf_mlp_bwd differentiates only the points whose d_rgbs is not four zeros (default; capi_train.cpp).  The
    reference's gradient of such a point is zero as well (nerf.py:236-239: alpha = 1 - exp(-relu(sigma) dist) gives a
    sample with sigma <= 0 no weight, tf.nn.relu no slope), so the sums are the sums over every point:
      * the device-side list is exactly np.nonzero of the upstream gradient, ascending, -0 counted as zero, NaN not;
      * the weight gradients equal the every-point call's (option nerf_bwd_rows = 0) to fp32 summation order;
      * two calls return the same bits; no point with a gradient: the gradients stay as they were."""
    from nerfactor_amd import ops
    from tests import common
    ks_np, bs_np = common.nerf_layers(common.nerf_nets(seed=5, opaque=False)[0])
    rng = np.random.default_rng(900 + s)
    rayo = rng.uniform(-1, 1, size=(n_rays, 3)).astype(np.float32)
    rayd = rng.normal(size=(n_rays, 3)).astype(np.float32)
    rayd /= np.linalg.norm(rayd, axis=1, keepdims=True)
    z = np.sort(rng.uniform(0.5, 3., size=(n_rays, s)).astype(np.float32), 1)
    n_pts = n_rays * s
    d_rgbs = rng.normal(size=(n_pts, 4)).astype(np.float32)
    on = rng.uniform(size=n_pts) < keep
    on[2048:5120] = False                      # whole 1024-point blocks without a gradient
    if keep > 0:
        on[7000:7300] = True
    d_rgbs[~on] = 0.
    d_rgbs[~on & (rng.uniform(size=n_pts) < 0.3)] = -0.     # signed zeros are zeros
    one = np.flatnonzero(on)[::7]
    d_rgbs[one, :3] = 0.                        # a gradient in one component is a gradient
    want_list = np.flatnonzero((d_rgbs != 0).any(1))
    blob = ops.pack_nerf_train_weights(ks_np, bs_np).to(cuda)
    args = (dev(rayo, cuda), dev(rayd, cuda), dev(z, cuda), dev(d_rgbs.reshape(n_rays, s, 4), cuda), blob)

    def run(rows):
        nfx_opt.set('nerf_bwd_rows', rows)
        dks = [torch.zeros(k.shape, device=cuda) for k in ks_np]
        dbs = [torch.zeros(b.shape, device=cuda) for b in bs_np]
        ws = ops.nerf_mlp_bwd(*args, dks, dbs)
        torch.cuda.synchronize()
        return torch.cat([t.reshape(-1) for t in dks + dbs]), ws

    dense, _ = run(0)
    listed, ws = run(1)
    words = ws.view(torch.int32)
    total = nfx_lib.lib.nfx_nerf_bwd_workspace_bytes(n_rays, s) // 4
    _, _, i0, n_words = ops.nerf_bwd_list_words(n_pts)    # [count, 3 pad][a count per 1024 points][indices], at the end
    tail = words[total - n_words:total].cpu().numpy()
    assert int(tail[0]) == want_list.size
    assert np.array_equal(tail[i0:i0 + want_list.size], want_list)
    assert torch.isfinite(listed).all()
    scale = float(dense.abs().max())
    if keep == 0.:
        assert scale == 0. and float(listed.abs().max()) == 0.
    else:
        assert scale > 0 and float((listed - dense).abs().max()) <= 2e-5 * scale
    assert torch.equal(run(1)[0], listed)
    # a NaN upstream is a gradient: it reaches the weights instead of being dropped with the zeros
    if keep == 1.0:
        bad = d_rgbs.copy()
        bad[:] = 0.
        bad[4321, 3] = np.nan
        args = args[:3] + (dev(bad.reshape(n_rays, s, 4), cuda), blob)
        assert not torch.isfinite(run(1)[0]).all()


def test_nerf_loss_as_one_kernel_equals_the_elementwise_chain(nfx_lib, cuda):
    """models/nerf.py compute_loss on device tensors = autograd.PairLoss (one forward, one backward launch); against the
    plain-torch chain of losses.L2 (nerf.py:292-300): per-ray values and both gradients."""
    from nerfactor_amd.nerfactor import losses
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    model = get_model_class('nerf')(make_config('nerf', loss='0.7l2')).to(cuda)
    rng = np.random.default_rng(44)
    n = 1000
    gt = dev(rng.uniform(size=(n, 3)).astype(np.float32), cuda)
    got, want = [], []
    for fused in (True, False):
        c = dev(rng.uniform(size=(n, 3)).astype(np.float32) if fused else got[1].detach().cpu().numpy(), cuda).requires_grad_()
        f = dev(rng.uniform(size=(n, 3)).astype(np.float32) if fused else got[2].detach().cpu().numpy(), cuda).requires_grad_()
        if fused:
            loss = model.compute_loss({'coarse': c, 'fine': f}, gt, keep_batch=True)
            assert type(loss.grad_fn).__name__.startswith('PairLoss')
        else:
            l2 = losses.L2()
            loss = 0.7 * l2(gt, c, keep_batch=True) + 0.7 * l2(gt, f, keep_batch=True)
        w = dev(np.linspace(0.5, 1.5, n).astype(np.float32), cuda)
        (loss * w).sum().backward()
        (got if fused else want).extend([loss.detach(), c, f])
    assert got[0].shape == (n,)
    np.testing.assert_allclose(got[0].cpu().numpy(), want[0].cpu().numpy(), rtol=1e-6, atol=1e-8)
    for a, b in ((got[1], want[1]), (got[2], want[2])):
        np.testing.assert_allclose(a.grad.cpu().numpy(), b.grad.cpu().numpy(), rtol=1e-6, atol=1e-9)


def test_nerf_train_step_descends(nfx_lib, cuda):
    """models.nerf through optim.train_step: every one of the 48 parameter tensors gets a finite gradient and the
    coarse + fine L2 loss (nerf.py:292-300) goes down on a fixed batch."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from tests import common
    torch.manual_seed(0)
    cfg = make_config('nerf', n_samples_coarse=16, n_samples_fine=32, perturb=False, lr='5e-4')
    model = get_model_class('nerf')(cfg).to(cuda)
    with torch.no_grad():   # some opacity so that the compositing gradient is not vanishing
        for pref in ('coarse_', 'fine_'):
            layer = model.net[pref + 'sigma_out'].layers[0]
            layer.kernel.mul_(8.)
            layer.bias.add_(0.5)
    rayo, rayd = common.camera_rays(16, 16)
    n = rayo.shape[0]
    rng = np.random.default_rng(3)
    rgb = rng.uniform(size=(n, 3)).astype(np.float32)
    batch = (['v'] * n, torch.tensor([[16, 16]] * n, dtype=torch.int32, device=cuda), dev(rayo, cuda),
             dev(rayd, cuda), dev(rgb, cuda))
    pred, gt, kw, _ = model(batch, mode='train')
    loss = model.compute_loss(pred, gt, keep_batch=True).sum() / n
    loss.backward()
    got = {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}
    assert len(got) == 48 and all(torch.isfinite(g).all() for g in got.values())
    opt = optim.make_optimizer(model, cfg)
    first = None
    for it in range(25):
        total, _ = optim.train_step(model, batch, opt, n)
        first = float(total) if first is None else first
    assert np.isfinite(float(total)) and float(total) < 0.9 * first, (first, float(total))
    assert abs(first - float(loss)) < 1e-5 * max(1., abs(first))
    # the cached device blobs follow the optimizer (its kernel writes the parameters through raw pointers)
    from nerfactor_amd import ops
    for pref in ('coarse_', 'fine_'):
        ks, bs = model._nerf_params(pref)
        assert torch.equal(model._nerf_blob(pref).cpu(), ops.pack_nerf_weights(ks, bs))
        assert torch.equal(model._nerf_train_blob(pref).cpu(), ops.pack_nerf_train_weights(ks, bs))


def test_packed_blobs_follow_optimizer_steps(nfx_lib, cuda):
    """Regression: after AMSGrad steps the width-128 blobs used by the kernels are the CURRENT weights
    (device re-pack through nfx_pack_gather, invalidated by the optimizer's version bump)."""
    from nerfactor_amd import ops, optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    torch.manual_seed(1)
    cfg = make_config('shape', lr='1e-2')
    model = get_model_class('shape')(cfg).to(cuda)
    rng, lxyz, _, xyz, cam, normal = scene(256, 17)
    lvis = rng.uniform(size=(256, 512)).astype(np.float32)
    batch = (['v'] * 256, torch.tensor([[16, 16]] * 256, dtype=torch.int32, device=cuda), dev(cam, cuda),
             dev(xyz - cam, cuda), dev(xyz * 0 + .5, cuda), torch.ones((256, 1), device=cuda), dev(xyz, cuda),
             dev(normal, cuda), dev(lvis, cuda))
    opt = optim.make_optimizer(model, cfg)
    before = model._blob128('normal_mlp', 'normal_out', nfx_lib.IN_XYZ, 3).clone()
    losses = [float(optim.train_step(model, batch, opt, 256)[0]) for _ in range(12)]
    assert losses[-1] < losses[0], losses
    for body, head, kind, od in (('normal_mlp', 'normal_out', nfx_lib.IN_XYZ, 3),
                                 ('lvis_mlp', 'lvis_out', nfx_lib.IN_XYZ_LDIR, 1)):
        params = model._params128(body, head)
        ks, bs = list(params[:5]), list(params[5:])
        assert torch.equal(model._blob128(body, head, kind, od).cpu(), ops.pack_mlp128_weights(ks, bs, kind, od))
        assert torch.equal(model._train_blob128(body, head, kind, od).cpu(),
                           ops.pack_mlp128_train_weights(ks, bs, kind, od))
    assert not torch.equal(before, model._blob128('normal_mlp', 'normal_out', nfx_lib.IN_XYZ, 3))


def test_shade_backward_finite_for_grazing_half_vectors(nfx_lib, cuda):
    """Regression: normals orthogonal (to ~1e-7) to the half vector of some light used to give inf - inf in the
    literal chain rule through cos^4 (a2 + tan^2)^2; the closed-form GGX derivatives stay finite."""
    from nerfactor_amd import ops
    n = 2048
    rng, lxyz, lareas, xyz, cam, normal = scene(n, 123)
    # make every point graze the half vector of light (i mod 512): n = unit vector orthogonal to h, + 1e-7 h
    v = cam - xyz
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    l = lxyz[np.arange(n) % 512] - xyz
    l /= np.linalg.norm(l, axis=1, keepdims=True)
    h = (l + v) / np.linalg.norm(l + v, axis=1, keepdims=True)
    t = np.cross(h, v)
    t /= np.linalg.norm(t, axis=1, keepdims=True)
    normal = (np.cos(0.3) * np.cross(t, h) + np.sin(0.3) * t + rng.choice([1e-7, 3e-8, 0., 1e-6], size=(n, 1)) * h)
    normal = normal.astype(np.float32)
    albedo = rng.uniform(.1, .8, size=(n, 3)).astype(np.float32)
    rough = rng.uniform(.1, .9, size=(n,)).astype(np.float32)
    lvis = rng.uniform(size=(n, 512)).astype(np.float32)
    light = rng.uniform(size=(512, 3)).astype(np.float32)
    drgb = rng.normal(size=(n, 3)).astype(np.float32)
    d_light = torch.zeros((512, 3), device=cuda)
    outs = ops.shade_bwd(dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(albedo, cuda), dev(lvis, cuda),
                         dev(lxyz, cuda), dev(lareas, cuda), dev(light, cuda), dev(drgb, cuda), d_light,
                         rough=dev(rough, cuda))
    for name, o in zip(('d_albedo', 'd_normal', 'd_lvis', 'd_rough'), outs):
        assert torch.isfinite(o).all(), name
    assert torch.isfinite(d_light).all()


def test_shade_backward_finite_for_vanishing_roughness_and_mirror_lights(nfx_lib, cuda):
    """Regression (round 3, found by the bench's own training leg: check_numerics raised at step 116 on the synthetic
    batch): a roughness that underflows (sigmoid output 7e-17 -> a2 = rough^4 = 0) with a half vector exactly along the
    normal (q = (h.n)^2 = 1) made the closed-form GGX derivatives 0 / 0; every quotient is a divide_no_nan now, as in the
    forward, and an underflowed roughness gets a zero gradient.  Normals are set to the half vector of light (i mod 512),
    the roughness covers 0, denormal-producing and ordinary values."""
    from nerfactor_amd import ops
    n = 2048
    rng, lxyz, lareas, xyz, cam, _ = scene(n, 321)
    v = cam - xyz
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    l = lxyz[np.arange(n) % 512] - xyz
    l /= np.linalg.norm(l, axis=1, keepdims=True)
    h = (l + v) / np.linalg.norm(l + v, axis=1, keepdims=True)
    normal = h.astype(np.float32)                        # mirror configuration for one light of every point
    albedo = rng.uniform(.1, .8, size=(n, 3)).astype(np.float32)
    rough = rng.choice(np.array([0., 6.7e-17, 1e-12, 1e-6, 3e-3, 0.2], np.float32), size=(n,)).astype(np.float32)
    lvis = rng.uniform(size=(n, 512)).astype(np.float32)
    light = rng.uniform(size=(512, 3)).astype(np.float32)
    drgb = rng.normal(size=(n, 3)).astype(np.float32)
    d_light = torch.zeros((512, 3), device=cuda)
    outs = ops.shade_bwd(dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(albedo, cuda), dev(lvis, cuda),
                         dev(lxyz, cuda), dev(lareas, cuda), dev(light, cuda), dev(drgb, cuda), d_light,
                         rough=dev(rough, cuda))
    for name, o in zip(('d_albedo', 'd_normal', 'd_lvis', 'd_rough'), outs):
        assert torch.isfinite(o).all(), (name, int((~torch.isfinite(o)).sum()))
    assert torch.isfinite(d_light).all()
    d_rough = outs[3].cpu().numpy()
    assert np.all(d_rough[rough < 1e-15] == 0.)          # rough^3 underflows (0 and 6.7e-17): no gradient
    rgb = ops.shade_fwd(dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(albedo, cuda), dev(lvis, cuda),
                        dev(lxyz, cuda), dev(lareas, cuda), dev(light[None], cuda), rough=dev(rough, cuda))
    assert torch.isfinite(rgb).all()


@pytest.mark.determinism
@pytest.mark.parametrize("wgrad_lds", ["0", "1", "fused"])
def test_weight_gradients_are_bit_reproducible(nfx_lib, cuda, nfx_opt, wgrad_lds):
    """No float atomics in the weight-gradient path: every (row slab, dW block) stores its partial sum and a second
    kernel adds the slabs in slab order, so two runs of the same backward give identical bits — for the direct-load
    kernel (short slabs in parallel) and for the LDS-staged one, for the width-128 and the NeRF networks; "fused" = the
    r04 default of the width-128 networks (accumulators in registers, per-workgroup slices added in workgroup order)."""
    from nerfactor_amd import ops
    if wgrad_lds == "fused":
        nfx_opt.set("wgrad_fused", 1)
    else:
        nfx_opt.set("wgrad_fused", 0)
        nfx_opt.set("wgrad_lds", wgrad_lds)
    layers, out = net128(31, 90, 1)
    ks = [k for k, _ in layers] + [out[0][0]]
    bs = [b for _, b in layers] + [out[0][1]]
    blob = ops.pack_mlp128_train_weights(ks, bs, nfx_lib.IN_XYZ_LDIR, 1).to(cuda)
    rng, lxyz, _, xyz, _, _ = scene(300, 77)
    dout = dev(rng.normal(size=(300, 512)), cuda)
    runs = []
    for _ in range(2):
        dks = [torch.zeros(k.shape, device=cuda) for k in ks]
        dbs = [torch.zeros(b.shape, device=cuda) for b in bs]
        ops.mlp128_bwd(nfx_lib.IN_XYZ_LDIR, dev(xyz, cuda), dout, blob, dks, dbs, out_act='sigmoid', lxyz=dev(lxyz, cuda))
        runs.append(dks + dbs)
    assert all(torch.equal(a, b) for a, b in zip(*runs))
    assert all(float(g.abs().max()) > 0 for g in runs[0])
    from tests import common
    net = common.nerf_nets(seed=3)[0]
    nks, nbs = common.nerf_layers(net)
    nblob = ops.pack_nerf_train_weights(nks, nbs).to(cuda)
    rayo, rayd = common.camera_rays(12, 12)
    o, d = dev(rayo, cuda), ops.l2_normalize3(dev(rayd, cuda), 1e-12)
    z = ops.gen_z(2., 6., 48, o.shape[0], device=cuda)
    d_rgbs = dev(rng.normal(size=(o.shape[0], 48, 4)), cuda)
    runs = []
    for _ in range(2):
        dks = [torch.zeros(k.shape, device=cuda) for k in nks]
        dbs = [torch.zeros(b.shape, device=cuda) for b in nbs]
        ops.nerf_mlp_bwd(o, d, z, d_rgbs, nblob, dks, dbs)
        runs.append(dks + dbs)
    assert all(torch.equal(a, b) for a, b in zip(*runs))


def test_numerics_verdicts_do_not_block_the_launch_queue(nfx_lib, cuda):
    """check_numerics under autograd leaves its verdict on the device; flush_numerics() ships it asynchronously and
    raises once the copy has landed (or at once with block=True) — the host never waits for the previous step."""
    from nerfactor_amd.nerfactor.models.base import Model

    class M(Model):
        def __init__(self):
            torch.nn.Module.__init__(self)
    m = M()
    good, bad = torch.ones(8, device=cuda), torch.tensor([1., float('inf')], device=cuda)
    m.check_numerics(good, "fine")
    m.flush_numerics()
    m.check_numerics(bad, "Normal")
    try:
        m.flush_numerics()                          # enqueued; raises here only if the copy has already landed
        landed = False
    except FloatingPointError as e:
        assert "Normal" in str(e)
        landed = True
    if not landed:
        torch.cuda.synchronize()
        with pytest.raises(FloatingPointError, match="Normal"):
            m.flush_numerics()                      # landed by now
    m.check_numerics(bad, "Loss")
    with pytest.raises(FloatingPointError, match="Loss"):
        m.flush_numerics(block=True)
    m.flush_numerics(block=True)


@pytest.mark.parametrize("kind", ["mae", "mse"])
def test_fused_pair_loss_vs_torch(nfx_lib, cuda, kind):
    """loss.hip against the torch formulation of the surface models' compute_loss (alpha blending as
    util/img.py:alpha_blend, keras MSE / MAE = mean over the last axis): values and every gradient, with a tensor that
    appears in two terms (its gradient accumulates), D = 3 / 512 / 1, background rays (alpha = 0) and ties."""
    from nerfactor_amd import autograd as nfx_grad
    from nerfactor_amd.nerfactor.util import img as imgutil
    g = torch.Generator(device='cpu').manual_seed(3)
    n = 1031
    mk = lambda d, grad: torch.rand(n, d, generator=g).to(cuda).requires_grad_(grad)
    rgb_p, rgb_g = mk(3, True), mk(3, False)
    lv_p, lv_g, lv_j = mk(512, True), mk(512, False), mk(512, True)
    r_p, r_j = mk(1, True), mk(1, True)
    with torch.no_grad():
        lv_j[:, :7] = lv_p[:, :7]            # exact ties: sign(0) = 0
    alpha = (torch.rand(n, 1, generator=g) > 0.3).float().to(cuda) * torch.rand(n, 1, generator=g).to(cuda)
    bg = 1.
    f = (lambda a, b: (a - b).abs().mean(-1)) if kind == 'mae' else (lambda a, b: ((a - b) ** 2).mean(-1))
    on_bg = lambda x: imgutil.alpha_blend(x, alpha, torch.full_like(x, bg))
    want = ((on_bg(rgb_g) - on_bg(rgb_p)) ** 2).mean(-1) + 0.1 * ((on_bg(lv_g) - on_bg(lv_p)) ** 2).mean(-1) \
        + 0.05 * f(on_bg(lv_p), lv_j) + 0.01 * f(r_p, r_j)
    up = torch.rand(n, generator=g).to(cuda)
    tensors = [rgb_p, lv_p, lv_j, r_p, r_j]
    want_g = torch.autograd.grad((want * up).sum(), tensors)
    spec = ((0, 1, 1., 'mse', True, True), (2, 3, 0.1, 'mse', True, True), (2, 4, 0.05, kind, True, False),
            (5, 6, 0.01, kind, False, False))
    got = nfx_grad.PairLoss.apply(alpha, bg, spec, rgb_p, rgb_g, lv_p, lv_g, lv_j, r_p, r_j)
    got_g = torch.autograd.grad((got * up).sum(), tensors)
    assert torch.allclose(got, want, rtol=2e-6, atol=1e-7)
    for a, b in zip(got_g, want_g):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-9), float((a - b).abs().max())
    # deterministic
    assert torch.equal(got, nfx_grad.PairLoss.apply(alpha, bg, spec, rgb_p, rgb_g, lv_p, lv_g, lv_j, r_p, r_j))


@pytest.mark.determinism
def test_graphed_nerf_train_step_equals_the_eager_one(nfx_lib, cuda):
    """The NeRF step (stratified coarse samples and inverse-CDF fine samples drawn with torch.rand INSIDE the step, perturb =
    True as nerf.ini trains) captured in a hipGraph: torch's generator hands every replay the draws the eager step would have
    made, so 40 steps on changing batches agree bit for bit with the eager run (the bench's NeRF training leg
    was timed eager)."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    n = 128
    rng = np.random.default_rng(3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)
    bs = []
    for i in range(4):
        cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
        bs.append((None, None, cam, t(rng.uniform(-1, 1, size=(n, 3))) - cam, t(rng.uniform(size=(n, 3)))))

    def run(graph):
        torch.manual_seed(11)
        cfg = make_config('nerf', n_samples_coarse='16', n_samples_fine='32')
        assert cfg.getboolean('DEFAULT', 'perturb')
        model = get_model_class('nerf')(cfg).to(cuda)
        opt = optim.make_optimizer(model, cfg)
        step = optim.GraphedTrainStep(model, opt, n, warmup=2) if graph else (lambda b: optim.train_step(model, b, opt, n))
        losses = [step(bs[i % len(bs)])[0] for i in range(40)]
        return torch.stack(losses), opt.flat.clone(), step
    le, pe, _ = run(False)
    lg, pg, step = run(True)
    assert len(step.graphs) == 1, "the step was not captured"
    assert torch.isfinite(le).all() and torch.equal(le, lg) and torch.equal(pe, pg)


@pytest.mark.determinism
@pytest.mark.parametrize("name,jitter,steps,precision", [
    ("shape", "0.01", 200, "bf16"), ("nerfactor_microfacet", "0.01", 200, "bf16"), ("nerfactor", "0.01", 200, "bf16"),
    ("nerfactor_microfacet", "0", 12, "bf16"), ("nerfactor_microfacet", "0.01", 12, "fp32")])
def test_graphed_train_step_equals_the_eager_one(nfx_lib, cuda, name, jitter, steps, precision):
    """optim.GraphedTrainStep (the step captured in a hipGraph and replayed) against optim.train_step: the two are the
    same kernels in the same order and torch's generator hands a replay the noise the eager step would have drawn, so
    losses, parameters and optimizer state after 200 steps on changing batches — xyz jitter ON, as the shipped configs
    train — must agree bit for bit and stay finite (round 2's six-step, jitter-off form missed that its benchmark ended
    in NaN; scripts/diag_graph_diverge.py is the step-by-step version of this test).  The version counters move, so an
    eager vali call afterwards sees the trained weights; the `to_vis` a replay returns belongs to the caller (clones of
    the graph's static outputs, `id` of the live batch).  precision = fp32: the step on the fp32 runtime-shaped kernels
    (workspaces from torch's allocator inside the capture) replays the same way."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
    from nerfactor_amd.nerfactor.models import get_model_class
    extra = dict(shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='') if name != 'shape' else {}
    n = 256

    def batches():
        rng = np.random.default_rng(7)
        out = []
        for i in range(6):
            t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)
            xyz = t(rng.uniform(-1, 1, size=(n, 3)))
            nrm = torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1)
            cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
            out.append((['view%d' % i] * n, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))),
                        mark_all_foreground(torch.ones(n, 1, device=cuda)), xyz, nrm, t(rng.uniform(size=(n, 512)))))
        return out

    def run(graph):
        torch.manual_seed(11)
        cfg = make_config(name, xyz_jitter_std=jitter, precision=precision, **extra)
        model = get_model_class(name)(cfg).to(cuda)
        opt = optim.make_optimizer(model, cfg)
        step = optim.GraphedTrainStep(model, opt, n, warmup=2) if graph else (lambda b: optim.train_step(model, b, opt, n))
        bs = batches()
        losses, kept = [], []
        for i in range(steps):
            loss, to_vis = step(bs[i % len(bs)])
            losses.append(loss)
            if i in (3, 4):           # replays (the capture happens at step 2): what trainvali keeps for an epoch
                kept.append(to_vis)
        model.flush_numerics(block=True)
        with torch.no_grad():
            vali = model(batches()[0], mode='vali')[0]
        return torch.stack(losses), opt.flat.clone(), opt.vhat.clone(), opt.iterations, vali, step, kept

    l0, p0, v0, it0, vali0, _, kept0 = run(False)
    l1, p1, v1, it1, vali1, step, kept1 = run(True)
    assert len(step.graphs) == 1 and it0 == it1 == steps
    assert bool(torch.isfinite(l0).all()) and bool(torch.isfinite(p0).all())
    differ = (l0 != l1).nonzero()[:, 0].tolist()   # every gradient is order-independent (ordered wgrad, fixed-point d_light)
    assert not differ, ("first differing step %d of %d" % (differ[0], steps), l0[differ[0]:differ[0] + 4], l1[differ[0]:differ[0] + 4])
    assert torch.equal(p0, p1) and torch.equal(v0, v1)
    for k in vali0:
        if isinstance(vali0[k], torch.Tensor):
            assert torch.equal(vali0[k], vali1[k]), k
    # the kept to_vis of two consecutive replays: each equals the eager step's and they differ from one another
    for a, b in zip(kept0, kept1):
        assert a['id'] == b['id']
        for k in a:
            if isinstance(a[k], torch.Tensor):
                assert torch.equal(a[k], b[k]), k
    assert kept1[0]['id'] != kept1[1]['id']
    key = 'pred_lvis' if 'pred_lvis' in kept1[0] else 'pred_normal'
    assert not torch.equal(kept1[0][key], kept1[1][key])


@pytest.mark.determinism
@pytest.mark.parametrize("name", ["nerfactor_microfacet", "nerfactor"])
def test_whole_step_gradients_are_bit_reproducible(nfx_lib, cuda, name):
    """No floating-point atomics are left in the training path (ordered weight-gradient reduction, fixed-point sums
    for the light gradient in shade_bwd and for d z / d normal in brdf_spec_bwd): the same step run twice from the
    same state gives identical gradients for every trainable tensor, the light included."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
    from nerfactor_amd.nerfactor.models import get_model_class
    torch.manual_seed(4)
    cfg = make_config(name, shape_mode='finetune', shape_model_ckpt='none', brdf_model_ckpt='none', test_envmap_dir='')
    model = get_model_class(name)(cfg).to(cuda)
    opt = optim.make_optimizer(model, cfg)
    rng = np.random.default_rng(9)
    n = 300
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)
    xyz = t(rng.uniform(-1, 1, size=(n, 3)))
    nrm = torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1)
    cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
    batch = (None, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))),
             mark_all_foreground(torch.ones(n, 1, device=cuda)), xyz, nrm, t(rng.uniform(size=(n, 512))))
    noise = t(rng.normal(size=(n, 3)) * 0.01)
    grads = []
    for _ in range(2):
        opt.zero_grad()
        pred, gt, kw, _ = model(batch, mode='train', xyz_noise=noise)
        (model.compute_loss(pred, gt, keep_batch=True, **kw).sum() / n).backward()
        grads.append(opt.bucket.flat.clone())
    assert torch.equal(grads[0], grads[1])
    assert float(model._light.grad.abs().max()) > 0


@pytest.mark.determinism
@pytest.mark.parametrize("name,n", [("nerf", 37), ("nerf", 1024), ("nerfactor_microfacet", 300), ("shape", 37)])
def test_ring_backward_kernels_equal_the_register_staged_ones(nfx_lib, cuda, nfx_opt, name, n):
    """The r03 backward kernels (weights through an LDS-DMA ring with hand-counted vmcnt waits; NeRF: 4 or 8 waves per
    workgroup) run the same MFMAs on the same operands as the register-staged kernels of rounds 1-2: every gradient of
    a whole step is bit-identical.  A wait that is one count too generous shows up here as a changed bit."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
    from nerfactor_amd.nerfactor.models import get_model_class
    torch.manual_seed(11)
    extra = dict(shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='') if 'nerfactor' in name else {}
    cfg = make_config(name, **extra)
    model = get_model_class(name)(cfg).to(cuda)
    opt = optim.make_optimizer(model, cfg)
    rng = np.random.default_rng(12)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)
    xyz = t(rng.uniform(-1, 1, size=(n, 3)))
    cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
    if name == 'nerf':
        batch = (None, None, cam, xyz - cam, t(rng.uniform(size=(n, 3))))
    else:
        batch = (None, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))),
                 mark_all_foreground(torch.ones(n, 1, device=cuda)), xyz,
                 torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1), t(rng.uniform(size=(n, 512))))

    nfx_opt.set('wgrad_fused', 0)   # the stored-activation path: the kernels this test compares (the fused kernels have their own)
    nfx_opt.set('nerf_bwd_rows', 0)  # every point, as the register-staged kernel has them (the row list: its own test below)

    def grads(**env):
        for k in ('nerf_bwd', 'nerf_bwd_nw', 'm128_bwd'):
            nfx_opt.unset(k)
        for k, v in env.items():
            nfx_opt.set(k, v)
        torch.manual_seed(13)   # the NeRF step draws its stratified samples and noise
        opt.zero_grad()
        pred, gt, kw, _ = model(batch, mode='train')
        (model.compute_loss(pred, gt, keep_batch=True, **kw).sum() / n).backward()
        return opt.bucket.flat.clone()

    ref = grads(nerf_bwd=0, m128_bwd=0)
    assert float(ref.abs().max()) > 0
    assert torch.equal(grads(), ref)                                     # the defaults: rings, NeRF with 8 waves
    if name == 'nerf':
        assert torch.equal(grads(nerf_bwd_nw=4), ref)
        if n >= 1024:   # over the rows with a gradient (another summation order than `ref`): 8 waves == 4 waves, run == run
            listed = grads(nerf_bwd_rows=1)
            assert torch.equal(grads(nerf_bwd_rows=1, nerf_bwd_nw=4), listed) and torch.equal(grads(nerf_bwd_rows=1), listed)
            assert float((listed - ref).abs().max()) <= 2e-5 * float(ref.abs().max())


@pytest.mark.determinism
def test_nerf_backward_chains_on_side_streams_equal_the_serial_ones(nfx_lib, cuda):
    """autograd.NerfMlp.backward puts the coarse and the fine network's backward (independent chains) on two side streams that
    the caller's stream joins when the backward pass ends: the gradients read right after loss.backward() — and after a
    hipGraph replay of the whole step — are bit for bit those of the two chains run one after the other."""
    from nerfactor_amd import autograd, optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    n = 1024
    rng = np.random.default_rng(21)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)
    cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
    batch = (None, None, cam, t(rng.uniform(-1, 1, size=(n, 3))) - cam, t(rng.uniform(size=(n, 3))))

    def run(side, graphed):
        torch.manual_seed(3)
        model = get_model_class('nerf')(make_config('nerf')).to(cuda)
        opt = optim.make_optimizer(model, model.config)
        prev, autograd.NERF_BWD_SIDE_STREAMS = autograd.NERF_BWD_SIDE_STREAMS, side
        try:
            torch.manual_seed(4)
            if graphed:
                step = optim.GraphedTrainStep(model, opt, n)
                for _ in range(4):
                    loss, _ = step(batch)
            else:
                for _ in range(4):
                    loss, _ = optim.train_step(model, batch, opt, n)
            torch.cuda.synchronize()
        finally:
            autograd.NERF_BWD_SIDE_STREAMS = prev
        assert not autograd._nerf_side['pending']
        return float(loss), opt.bucket.flat.clone(), torch.cat([p.detach().reshape(-1) for p in model.parameters()])

    serial = run(False, False)
    assert float(serial[1].abs().max()) > 0
    assert autograd.NERF_BWD_SIDE_STREAMS == 'capture'      # the default: forked inside a capture only
    for side, graphed in ((True, False), (True, True), ('capture', True), (False, True)):
        got = run(side, graphed)
        assert got[0] == serial[0] and torch.equal(got[1], serial[1]) and torch.equal(got[2], serial[2]), (side, graphed)


@pytest.mark.parametrize("name", ["nerfactor_microfacet", "nerfactor", "nerf", "shape"])
def test_precision_fp32_trains(nfx_lib, cuda, name):
    """`precision = fp32` (the reference computes in fp32, trainvali.py:110-127): the forward runs the fp32-class kernels
    (bf16 hi / lo operand pairs), the backward the bf16-operand kernels (autograd.GRAD_PREC) — a run no longer stops at its
    first backward call.  Checked: the step's loss equals the fp32-class forward's (not the bf16 one's), eight AMSGrad
    steps on one batch are finite and bring the loss down, and the gradient of step 1 agrees with the all-bf16 step's
    within bf16 operand noise (relative Frobenius <= 5 %)."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
    from nerfactor_amd.nerfactor.models import get_model_class
    n = 192
    rng = np.random.default_rng(3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)
    xyz = t(rng.uniform(-1, 1, size=(n, 3)))
    cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
    if name == 'nerf':
        batch = (None, None, cam, xyz - cam, t(rng.uniform(size=(n, 3))))
    else:
        batch = (None, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))),
                 mark_all_foreground(torch.ones(n, 1, device=cuda)), xyz,
                 torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1), t(rng.uniform(size=(n, 512))))
    extra = dict(shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='') if 'nerfactor' in name else {}
    if name == 'nerf':
        extra = dict(perturb='False')

    def run(prec, steps):
        torch.manual_seed(21)
        cfg = make_config(name, precision=prec, xyz_jitter_std='0', **extra)
        model = get_model_class(name)(cfg).to(cuda)
        opt = optim.make_optimizer(model, cfg)
        losses, grad = [], None
        for i in range(steps):
            losses.append(float(optim.train_step(model, batch, opt, n)[0]))
            if i == 0:
                grad = opt.bucket.flat[:-1].clone()
        model.flush_numerics(block=True)
        return losses, grad

    l32, g32 = run('fp32', 8)
    l16, g16 = run('bf16', 1)
    assert all(np.isfinite(l32)) and l32[-1] < l32[0], l32
    rel = float((g32 - g16).norm() / g16.norm())
    assert rel < 5e-2, rel
    assert abs(l32[0] - l16[0]) < 2e-2 * abs(l16[0]) + 1e-4, (l32[0], l16[0])
    print(name, "fp32-class training: loss %.5f -> %.5f (bf16 first loss %.5f), step-1 gradient vs bf16 step: %.2e" % (
        l32[0], l32[-1], l16[0], rel))
