"""The runtime-shaped kernels (csrc/mlp_generic.hip: nfx_mlp_generic_fwd, nfx_embed) against the oracle, and the NeRF
plugin on shapes the tuned kernels do not cover (VERDICT r03 missing #2; reference nerfactor/models/nerf.py:53-90,
nerfactor/networks/mlp.py:24-50): other widths and depths, skip anywhere, use_views = False, pos_enc = False."""
import numpy as np
import pytest
import torch

from oracle import nerf_ref
from tests import common

pytestmark = pytest.mark.gpu


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)


@pytest.mark.parametrize("d_in,widths,acts,skip_at,n", [
    (63, [96, 96, 96, 96, 5], ['relu'] * 4 + [None], [1], 1000),
    (3, [64, 64, 4], ['relu', 'relu', None], None, 77),
    (90, [256] * 8 + [1], ['relu'] * 8 + ['sigmoid'], [4], 333),
    (27, [40, 200, 33], ['relu', 'softplus', None], [0, 1], 64),
    (128, [256], ['relu'], None, 1), (283, [128, 3], ['relu', None], None, 200),
    # round 5: widths up to 512 (9 .. 16 output tiles per layer: one wave per workgroup), a 512-wide NeRF's colour head
    (63, [512, 512, 288, 512, 4], ['relu'] * 4 + [None], [1], 300), (539, [256, 3], ['relu', 'sigmoid'], None, 150)])
@pytest.mark.parametrize("prec", ['bf16', 'fp32', 'fp32_native'])
def test_generic_mlp_vs_oracle(nfx_lib, cuda, d_in, widths, acts, skip_at, n, prec):
    """bf16: same-rounding bound 4e-3, float64 bound 3e-2; prec = 'fp32_native' (fp32 operands, native fp32 matrix
    instruction): 1e-5 of the float64 network, relative to its largest output; prec = 'fp32' (fp32 activations, bf16
    hi / lo operand pairs = 16 significant bits per operand, round 5): 5e-5, the bound of the tuned fp32-class kernels."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(sum(widths))
    layers, prev = [], d_in
    for i, w in enumerate(widths):
        layers.append((nerf_ref.glorot_uniform(rng, prev, w), rng.uniform(-.2, .2, size=w).astype(np.float32)))
        prev = w + (d_in if skip_at and i in skip_at else 0)
    if skip_at and (len(widths) - 1) in skip_at:
        pytest.skip("a skip behind the last layer changes the output width")
    net = ops.GenericNet([k for k, _ in layers], [b for _, b in layers], acts, skip_at, prec=prec).to(cuda)
    x = rng.normal(size=(n, d_in)).astype(np.float32)
    got = ops.mlp_generic_fwd(dev(x, cuda), net).cpu().numpy()
    want = nerf_ref.mlp(x.astype(np.float64), [(k.astype(np.float64), b.astype(np.float64)) for k, b in layers], acts, skip_at)
    want_q = nerf_ref.mlp(x, layers, acts, skip_at, quant=nerf_ref.bf16_round)
    scale = max(1., np.abs(want).max())
    assert got.shape == (n, widths[-1])
    if prec != 'bf16':
        assert np.abs(got - want).max() < (1e-5 if prec == 'fp32_native' else 5e-5) * scale, np.abs(got - want).max()
    else:
        assert np.abs(got - want_q).max() < 4e-3 * scale, np.abs(got - want_q).max()      # same bf16 operand rounding
        assert np.abs(got - want).max() < 3e-2 * scale
    # writing into a column range of a wider matrix; a strided input
    wide = torch.full((n, widths[-1] + 7), -5., device=cuda)
    xs = torch.zeros((n, d_in + 3), device=cuda)
    xs[:, :d_in] = dev(x, cuda)
    ops.mlp_generic_fwd(xs, net, out=wide, col0=4)
    assert torch.equal(wide[:, 4:4 + widths[-1]].cpu(), torch.from_numpy(got))
    assert bool((wide[:, :4] == -5).all()) and bool((wide[:, 4 + widths[-1]:] == -5).all())
    assert ops.mlp_generic_fwd(torch.zeros((0, d_in), device=cuda), net).shape == (0, widths[-1])


def test_embed_kernel_is_the_embedder(nfx_lib, cuda):
    from nerfactor_amd import ops
    rng = np.random.default_rng(3)
    x = rng.uniform(-3, 3, size=(500, 3)).astype(np.float32)
    for L in (0, 2, 4, 10):
        got = ops.embed(L, x=dev(x, cuda)).cpu().numpy()
        want = nerf_ref.embed(x.astype(np.float64), L)
        assert got.shape == want.shape and np.abs(got - want).max() < 4e-6 * max(1, 2 ** L) / 16 + 2e-6
    o = rng.uniform(-1, 1, size=(40, 3)).astype(np.float32)
    d = nerf_ref.l2_normalize(rng.normal(size=(40, 3)).astype(np.float32), 1, 1e-12)
    z = np.sort(rng.uniform(2, 6, size=(40, 7)).astype(np.float32), 1)
    pts = (o[:, None] + d[:, None] * z[:, :, None]).reshape(-1, 3)
    got = ops.embed(10, rayo=dev(o, cuda), rayd=dev(d, cuda), z=dev(z, cuda)).cpu().numpy()
    assert np.abs(got - nerf_ref.embed(pts.astype(np.float64), 10)).max() < 5e-4     # fp32 points, bands up to 2^9
    wide = torch.zeros((40 * 7, 40), device=cuda)
    ops.embed(4, rayd=dev(d, cuda), per_ray=7, out=wide, col0=13)
    want = nerf_ref.embed(np.repeat(d, 7, 0).astype(np.float64), 4)
    assert np.abs(wide[:, 13:].cpu().numpy() - want).max() < 2e-6 and bool((wide[:, :13] == 0).all())


@pytest.mark.parametrize("overrides,kw", [
    (dict(mlp_width='128', enc_depth='4'), dict(width=128, depth=4)),
    (dict(mlp_width='64', enc_depth='6', n_freqs_xyz='6', n_freqs_view='2'), dict(width=64, depth=6, n_freqs_xyz=6, n_freqs_view=2)),
    (dict(use_views='False'), dict(use_views=False, n_freqs_view=0)),
    (dict(pos_enc='False', mlp_width='128'), dict(width=128, n_freqs_xyz=0, n_freqs_view=0)),
    # enc_depth = 2: skip_at = [1] = the LAST encoder layer, every head reads concat(y, embed(x)) (nerf.py:53-71)
    (dict(enc_depth='2', mlp_width='64'), dict(width=64, depth=2)),
    # round 5: mlp_width up to 512 (the colour head reads concat(512 bottleneck features, embedded view) = 539 inputs)
    # (wider layers sum more bf16-rounded products: the bound of these two random-weight renders is 5e-2; the kernels themselves
    #  are held to the same-rounding bound 4e-3 and, fp32-class, to 5e-5 at these widths: test_generic_mlp_vs_oracle)
    (dict(mlp_width='512', enc_depth='4'), dict(width=512, depth=4, max_abs=5e-2)),
    (dict(mlp_width='384'), dict(width=384, max_abs=5e-2))])      # (3.8e-2 measured: eight 384-wide layers of bf16-rounded products)
def test_nerf_plugin_renders_non_shipped_shapes(nfx_lib, cuda, overrides, kw):
    """Model.call(mode='test') of NeRF configurations outside config/nerf.ini's architecture against the oracle's render
    of the same weights: the stated tolerance (max-abs 3e-2 outside the alpha_last band, PSNR >= 40 dB)."""
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    cfg = make_config('nerf', **overrides)
    model = get_model_class('nerf')(cfg).to(cuda)
    assert not model.tuned
    rng = np.random.default_rng(5)
    lx, lv = kw.get('n_freqs_xyz', 10), kw.get('n_freqs_view', 4)
    nets = []
    for pref in ('coarse_', 'fine_'):
        net = nerf_ref.init_nerf_net(rng, n_freqs_xyz=lx, n_freqs_view=lv, width=kw.get('width', 256), depth=kw.get('depth', 8),
                                     sigma_bias=0.5, sigma_gain=8., use_views=kw.get('use_views', True))
        nerf_ref.randomize_biases(net, rng)
        nets.append(net)
        for part, pairs in net.items():
            for layer, (k, b) in zip(model.net[pref + part].layers, pairs):
                with torch.no_grad():
                    layer.kernel.copy_(torch.from_numpy(k))
                    layer.bias.copy_(torch.from_numpy(b))
    rayo, rayd = common.camera_rays(12, 12)
    n = rayo.shape[0]
    batch = (['v'] * n, torch.tensor([[12, 12]] * n), dev(rayo, cuda), dev(rayd, cuda), torch.rand(n, 3, device=cuda))
    pred, gt, loss_kwargs, to_vis = model(batch, mode='test')
    coarse, fine, aux = nerf_ref.render_rays(rayo, rayd, nets[0], nets[1], n_freqs_xyz=lx, n_freqs_view=lv)
    ok = (np.abs(aux['rgbs_coarse'][:, -1, 3]) > 0.1) & (np.abs(aux['rgbs_fine'][:, -1, 3]) > 0.1)   # (no fp32-class last sample here)
    assert ok.mean() > 0.6
    for tag, ref in (('coarse', coarse), ('fine', fine)):
        got = pred[tag].cpu().numpy()
        err = np.abs(got - ref['rgb']).max(-1)
        assert err[ok].max() <= kw.get('max_abs', 3e-2), (tag, err[ok].max())
        # (rays inside the alpha_last band flip without the tuned path's fp32-class last sample: PSNR over the stable rays)
        assert nerf_ref.psnr_uint8_luma(got[ok].reshape(-1, 1, 3), ref['rgb'][ok].reshape(-1, 1, 3)) >= 40.


@pytest.mark.parametrize("overrides,width,depth,skip,lx,ll", [
    (dict(mlp_width='64', mlp_depth='3', mlp_skip_at='1'), 64, 3, 1, 10, 4),
    (dict(n_freqs_xyz='6', n_freqs_ldir='2'), 128, 4, 2, 6, 2),
    (dict(mlp_width='256', mlp_depth='6', mlp_skip_at='3', xyz_scale='0.5'), 256, 6, 3, 10, 4),
    (dict(pos_enc='False', mlp_width='64'), 64, 4, 2, 0, 0),            # tf.identity embedders (shape.py:97-106), VERDICT r04 #8
    (dict(mlp_depth='3', mlp_skip_at='2'), 128, 3, 2, 10, 4),           # the skip behind the body's last layer: the head reads concat(y, x)
    (dict(mlp_width='320', mlp_depth='3', mlp_skip_at='1'), 320, 3, 1, 10, 4)])   # round 5: more than 8 output tiles per layer
def test_shape_plugin_renders_non_shipped_shapes(nfx_lib, cuda, overrides, width, depth, skip, lx, ll):
    """Surface MLPs outside mlp_width = 128 / mlp_depth = 4 / mlp_skip_at = 2 / bands 10, 4 (reference shape.py:79-94 builds
    them from the ini): normals and the [points x 512 lights] visibilities of Model.call(mode='test') against the oracle
    (bf16 bound 3e-2, same-rounding bound 1e-2)."""
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from oracle import nerfactor_ref as R
    cfg = make_config('shape', xyz_jitter_std='0', **overrides)
    model = get_model_class('shape')(cfg).to(cuda)
    assert not model._net_tuned('lvis_mlp')
    rng = np.random.default_rng(8)
    net = {}
    for name, d_in, d_out in (('normal', 3 + 6 * lx, 3), ('lvis', 6 + 6 * lx + 6 * ll, 1)):
        layers, out = R.init_mlp128(rng, d_in, d_out, width=width, depth=depth, skip_at=skip)
        for lst in (layers, out):
            for i, (k, b) in enumerate(lst):
                lst[i] = (k, rng.uniform(-.2, .2, size=b.shape).astype(np.float32))
        net[name + '_mlp'], net[name + '_out'] = layers, out
        for part, pairs in ((name + '_mlp', layers), (name + '_out', out)):
            for layer, (k, b) in zip(model.net[part].layers, pairs):
                with torch.no_grad():
                    layer.kernel.copy_(torch.from_numpy(k))
                    layer.bias.copy_(torch.from_numpy(b))
    n = 150
    xyz = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    lxyz = model.lxyz.reshape(-1, 3).cpu().numpy()
    z3 = np.zeros((n, 3), np.float32)
    batch = (['x'] * n, torch.tensor([[1, n]] * n), dev(z3, cuda), dev(z3, cuda), dev(z3, cuda),
             torch.ones(n, 1, device=cuda), dev(xyz, cuda), dev(z3 + 1, cuda), torch.zeros(n, lxyz.shape[0], device=cuda))
    pred, gt, kw, _ = model(batch, mode='test')
    scale = float(overrides.get('xyz_scale', 1.))
    surf2l = R.calc_ldir(xyz, lxyz)
    want_n = R.pred_normal_at(xyz, net, xyz_scale=scale, n_freqs_xyz=lx, skip_at=skip)
    want_n = want_n / np.maximum(np.linalg.norm(want_n, axis=1, keepdims=True), 1e-3)
    assert np.abs(pred['normal'].cpu().numpy() - want_n).max() < 3e-2
    want_l = R.pred_lvis_at(xyz, surf2l, net, xyz_scale=scale, n_freqs_xyz=lx, n_freqs_ldir=ll, skip_at=skip)
    want_q = R.pred_lvis_at(xyz, surf2l, net, xyz_scale=scale, quant=nerf_ref.bf16_round, n_freqs_xyz=lx, n_freqs_ldir=ll, skip_at=skip)
    got_l = pred['lvis'].cpu().numpy()
    assert got_l.shape == want_l.shape and np.abs(got_l - want_l).max() < 3e-2 and np.abs(got_l - want_q).max() < 1e-2


# ---------------------------------------------------------------------------------------------------- backward
def _rel(got, want):
    return float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-30))


def _oracle_grads(x, layers, acts, skip_at, dy, want_dx, quant=True):
    """d sum(y * dy) / d (kernels, biases, x) by torch.autograd over oracle/torch_train_ref.mlp in float64, operands of
    every Dense layer rounded to bf16 with a straight-through gradient (what the MFMA path computes)."""
    from oracle import torch_train_ref as T
    P = {}
    for i, (k, b) in enumerate(layers):
        P['net_m_layer%d.kernel' % i] = torch.tensor(k, dtype=torch.float64, requires_grad=True)
        P['net_m_layer%d.bias' % i] = torch.tensor(b, dtype=torch.float64, requires_grad=True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=want_dx)
    T.QUANT = {True: T.bf16_ste, 'bf16': T.bf16_ste, 'pairs': T.pairs_ste, False: None, None: None}[quant]
    try:
        y = T.mlp(xt, P, 'm', len(layers), acts, skip_at)
    finally:
        T.QUANT = None
    (y * torch.tensor(dy, dtype=torch.float64)).sum().backward()
    n = len(layers)
    return ([P['net_m_layer%d.kernel' % i].grad.numpy() for i in range(n)],
            [P['net_m_layer%d.bias' % i].grad.numpy() for i in range(n)], xt.grad.numpy() if want_dx else None)


@pytest.mark.parametrize("d_in,widths,acts,skip_at,n", [
    (63, [96, 96, 96, 96, 5], ['relu'] * 4 + [None], [1], 1000),
    (3, [64, 64, 4], ['relu', 'relu', None], None, 77),
    (90, [256] * 8 + [1], ['relu'] * 8 + ['sigmoid'], [4], 3333),
    (27, [40, 200, 33], ['relu', 'softplus', None], [0, 1], 64),
    (128, [256], ['relu'], None, 1), (283, [128, 3], ['relu', None], None, 200),
    (39, [128, 128, 128, 1], ['relu'] * 3 + ['sigmoid'], [1], 20000),
    (63, [512, 320, 512, 2], ['relu'] * 3 + [None], [1], 1500), (539, [256, 3], ['relu', 'sigmoid'], None, 150)])
@pytest.mark.parametrize("prec", ['bf16', 'fp32', 'fp32_native'])
def test_generic_mlp_backward_vs_oracle(nfx_lib, cuda, d_in, widths, acts, skip_at, n, prec):
    """nfx_mlp_generic_bwd: weight, bias and input gradients of arbitrary mlp.Network shapes against torch.autograd of
    the oracle with the same bf16 operand rounding.  Bound: 2 % of each tensor's norm — the kernels round the
    propagated gradient and the transposed weights to bf16 as well (two more 2^-9 roundings per layer), the oracle's
    straight-through backward does not.  prec = 'fp32_native': against the PLAIN float64 autograd, 2e-5 of each tensor's
    norm (nothing is rounded to bf16; the sums are fp32); prec = 'fp32' (hi / lo operand pairs in all three products —
    forward, dgrad, weight gradients): 2e-4, a hundredth of the bf16 bound, against the float64 network on the same
    16-bit operands.  Both fp32 modes are compared on the rows whose ReLU masks do not hang on the last bits (below).
    Bit-identical between calls; ADDS into the gradient buffers."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(sum(widths) + n)
    layers, prev = [], d_in
    for i, w in enumerate(widths):
        layers.append((nerf_ref.glorot_uniform(rng, prev, w), rng.uniform(-.2, .2, size=w).astype(np.float32)))
        prev = w + (d_in if skip_at and i in skip_at else 0)
    net = ops.GenericNet([k for k, _ in layers], [b for _, b in layers], acts, skip_at, train=True, prec=prec).to(cuda)
    x = rng.normal(size=(n, d_in)).astype(np.float32)
    dy_all = rng.normal(size=(n, widths[-1])).astype(np.float32)
    dy = dy_all
    if prec != 'bf16' and n >= 1000:
        # A pre-activation within rounding of 0 decides a ReLU mask, and a mask that differs from the oracle's moves its
        # row's whole path: with pre-activations ~1e-6 (pairs) / ~1e-7 (native) from float64's, a few of the 1e7 (row, unit)
        # masks of a large batch differ and the relative Frobenius distance of a weight gradient is 2e-3 ... 9e-3 (r05 calls
        # A / B: 128 x 3 on 20 000 rows, 256 x 8 on 3 333) however exact the products are.  The arithmetic is therefore
        # compared on the rows with no pre-activation inside 1e-4 of its layer's scale (dy = 0 elsewhere), every row
        # loosely at the end.
        h, x64, fragile = x.astype(np.float64), x.astype(np.float64), np.zeros(n, bool)
        for i, (k, b) in enumerate(layers):
            zpre = h @ k.astype(np.float64) + b
            if acts[i] == 'relu':
                fragile |= (np.abs(zpre) < 1e-4 * np.sqrt((zpre ** 2).mean())).any(1)
            h = {'relu': lambda t: np.maximum(t, 0), 'sigmoid': lambda t: 1. / (1. + np.exp(-t)), 'softplus': lambda t: np.logaddexp(t, 0),
                 None: lambda t: t}[acts[i]](zpre)
            if skip_at and i in skip_at:
                h = np.concatenate([h, x64], 1)
        assert fragile.mean() < 0.5
        dy = np.where(fragile[:, None], 0., dy_all).astype(np.float32)
    # the train blob's head is the forward blob
    fwd = ops.GenericNet([k for k, _ in layers], [b for _, b in layers], acts, skip_at, prec=prec).to(cuda)
    assert torch.equal(ops.mlp_generic_fwd(dev(x, cuda), net), ops.mlp_generic_fwd(dev(x, cuda), fwd))

    def run(fill, dy=dy):
        dks = [torch.full(k.shape, fill, device=cuda) for k, _ in layers]
        dbs = [torch.full(b.shape, fill, device=cuda) for _, b in layers]
        dx = ops.mlp_generic_bwd(dev(x, cuda), net, dev(dy, cuda), dks, dbs, want_dx=True)
        return dks, dbs, dx
    dks, dbs, dx = run(0.)
    wk, wb, wx = _oracle_grads(x, layers, acts, skip_at, dy, True, quant={'bf16': 'bf16', 'fp32': 'pairs', 'fp32_native': None}[prec])
    tol = 2e-2 if prec == 'bf16' else ((2e-5 if prec == 'fp32_native' else 2e-4) if len(widths) <= 5 else (5e-4 if prec == 'fp32_native' else 1e-3))
    if prec != 'bf16' and n >= 1000:      # every row, plain float64: the ReLU-mask effect included
        ak, ab, ax = run(0., dy_all)
        pk, pb, px = _oracle_grads(x, layers, acts, skip_at, dy_all, True, quant=None)
        worst = max(_rel(got.cpu().numpy(), want) for got, want in list(zip(ak, pk)) + list(zip(ab, pb)) + [(ax, px)])
        print("generic bwd %s, %d rows x %s: all rows vs plain float64, worst tensor %.2e" % (prec, n, widths, worst))
        assert worst < 3e-2, worst
    for i in range(len(layers)):
        assert _rel(dks[i].cpu().numpy(), wk[i]) < tol, ('kernel', i, _rel(dks[i].cpu().numpy(), wk[i]))
        assert _rel(dbs[i].cpu().numpy(), wb[i]) < tol, ('bias', i, _rel(dbs[i].cpu().numpy(), wb[i]))
    assert dx.shape == (n, d_in) and _rel(dx.cpu().numpy(), wx) < tol, _rel(dx.cpu().numpy(), wx)
    dks2, dbs2, dx2 = run(1.)
    for a, b in zip(dks + dbs, dks2 + dbs2):
        assert torch.allclose(a + 1., b, rtol=0, atol=2e-6 * max(1., float(a.abs().max())))     # accumulates
    dks3, dbs3, dx3 = run(0.)
    for a, b in zip(dks + dbs + [dx], dks3 + dbs3 + [dx3]):
        assert torch.equal(a, b)                                                               # deterministic
    # no input gradient requested: the same weight gradients
    dks4 = [torch.zeros_like(t) for t in dks]
    assert ops.mlp_generic_bwd(dev(x, cuda), net, dev(dy, cuda), dks4, [torch.zeros_like(t) for t in dbs]) is None
    for a, b in zip(dks, dks4):
        assert torch.equal(a, b)
    with pytest.raises(Exception, match='train'):
        ops.mlp_generic_bwd(dev(x, cuda), fwd, dev(dy, cuda), dks4, dbs)


def test_nerf_plugin_trains_non_shipped_shapes(nfx_lib, cuda):
    """models.nerf.Model of a non-shipped shape under autograd (one GenericMlp node per network, Composite, the l2
    loss): the first step's gradients against torch.autograd of oracle/torch_train_ref.nerf_loss with the same uniform
    draws, then 40 AMSGrad steps on the fixed batch reduce the loss."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from oracle import torch_train_ref as T
    nc, nf = 16, 32
    cfg = make_config('nerf', mlp_width='64', enc_depth='4', n_freqs_xyz='6', n_freqs_view='2',
                      n_samples_coarse=str(nc), n_samples_fine=str(nf), lr='1e-3')
    assert cfg.getboolean('DEFAULT', 'perturb') and cfg.getfloat('DEFAULT', 'noise_std') == 0.
    torch.manual_seed(11)
    model = get_model_class('nerf')(cfg)
    rng = np.random.default_rng(12)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('.bias'):
                p.copy_(torch.from_numpy(rng.uniform(-.1, .1, size=tuple(p.shape)).astype(np.float32)))
            if 'sigma_out' in name and name.endswith('.kernel'):
                p.mul_(8.)
    model = model.to(cuda)
    assert not model.tuned
    model.register_trainable()
    rayo, rayd = common.camera_rays(16, 16)
    n = rayo.shape[0]
    gt = rng.uniform(0, 1, size=(n, 3)).astype(np.float32)
    batch = (['x'] * n, torch.tensor([[16, 16]] * n), dev(rayo, cuda), dev(rayd, cuda), dev(gt, cuda))
    u_c, u_f = rng.uniform(size=(n, nc)).astype(np.float32), rng.uniform(size=(n, nf)).astype(np.float32)
    draws = iter([u_c, u_f])
    real_rand = torch.rand

    def replay(shape, device=None, **kw):
        a = next(draws)
        assert tuple(a.shape) == tuple(shape)
        return torch.from_numpy(a).to(device)
    torch.rand = replay
    try:
        pred, gt_, loss_kwargs, _ = model(batch, mode='train')
        loss = model.compute_loss(pred, gt_, keep_batch=True).sum() / n
        loss.backward()
    finally:
        torch.rand = real_rand
    got = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.requires_grad}
    P = {k: p.detach().cpu().double().requires_grad_(True) for k, p in model.named_parameters()}
    t64 = lambda a: torch.from_numpy(np.asarray(a, np.float64))
    T.QUANT = T.bf16_ste
    try:
        ref = T.nerf_loss(P, t64(rayo), t64(rayd), t64(gt), t64(u_c), torch.zeros(n, nc, dtype=torch.float64), t64(u_f),
                          torch.zeros(n, nc + nf, dtype=torch.float64), near=cfg.getfloat('DEFAULT', 'near'),
                          far=cfg.getfloat('DEFAULT', 'far'), n_coarse=nc, n_fine=nf,
                          white_bg=cfg.getboolean('DEFAULT', 'white_bg'), arch=(6, 2, 4, True)).sum() / n
    finally:
        T.QUANT = None
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 2e-2 * float(ref.detach()), (float(loss.detach()), float(ref.detach()))
    assert set(got) == set(P)
    worst = max((_rel(got[k], P[k].grad.numpy()), k) for k in got)
    assert worst[0] < 5e-2, worst
    opt = optim.make_optimizer(model, cfg)
    losses = []
    for step in range(40):
        opt.zero_grad()
        pred, gt_, _, _ = model(batch, mode='train')
        l = model.compute_loss(pred, gt_, keep_batch=True).sum() / n
        l.backward()
        losses.append(float(opt.step(loss=l.detach())))
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.8 * np.mean(losses[:5]), (losses[:5], losses[-5:])


def test_shape_plugin_trains_non_shipped_shapes(nfx_lib, cuda):
    """models.shape.Model with 64-wide, 3-deep surface MLPs: the gradients of one training call against
    torch.autograd of the oracle's networks (float64, bf16 operand rounding), then the loss falls under AMSGrad."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from oracle import torch_train_ref as T
    cfg = make_config('shape', xyz_jitter_std='0', mlp_width='64', mlp_depth='3', mlp_skip_at='1', lr='1e-3')
    torch.manual_seed(3)
    model = get_model_class('shape')(cfg).to(cuda)
    assert not model._net_tuned('normal_mlp')
    model.register_trainable()
    rng = np.random.default_rng(4)
    n = 96
    xyz = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    normal = nerf_ref.l2_normalize(rng.normal(size=(n, 3)).astype(np.float32), 1, 1e-12)
    L = model.lxyz.reshape(-1, 3).shape[0]
    lvis = rng.uniform(size=(n, L)).astype(np.float32)
    z3 = np.zeros((n, 3), np.float32)
    batch = (['x'] * n, torch.tensor([[1, n]] * n), dev(z3, cuda), dev(z3, cuda), dev(z3, cuda),
             torch.ones(n, 1, device=cuda), dev(xyz, cuda), dev(normal, cuda), dev(lvis, cuda))
    pred, gt, kw, _ = model(batch, mode='train')
    loss = model.compute_loss(pred, gt, **kw).mean()
    loss.backward()
    got = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.requires_grad}
    assert len(got) == 16 and all(np.abs(g).max() > 0 for g in got.values())
    # oracle: the same two networks and loss (shape.py:239-277 with alpha = 1: MSE over the last axis of both heads)
    P = {k: p.detach().cpu().double().requires_grad_(True) for k, p in model.named_parameters()}
    lxyz = model.lxyz.reshape(-1, 3).cpu().double()
    x64 = torch.from_numpy(xyz.astype(np.float64)) * model.xyz_scale
    T.QUANT = T.bf16_ste
    try:
        pe = T.embed(x64, 10)
        nrm = T.mlp(T.mlp(pe, P, 'normal_mlp', 3, ['relu'] * 3, [1]), P, 'normal_out', 1, [None]) + 1e-6
        nrm = T.l2n(nrm, 1, 1e-6)
        d = lxyz[None] - torch.from_numpy(xyz.astype(np.float64))[:, None]
        le = T.embed(T.l2n(d, 2, 1e-6), 4)
        rows = torch.cat((pe[:, None].expand(n, L, pe.shape[1]), le), -1).reshape(n * L, -1)
        vis = T.mlp(T.mlp(rows, P, 'lvis_mlp', 3, ['relu'] * 3, [1]), P, 'lvis_out', 1, ['sigmoid']).reshape(n, L)
    finally:
        T.QUANT = None
    w_n, w_l = cfg.getfloat('DEFAULT', 'normal_loss_weight'), cfg.getfloat('DEFAULT', 'lvis_loss_weight')
    ref = (w_n * ((nrm - torch.from_numpy(normal.astype(np.float64))) ** 2).mean(-1) +
           w_l * ((vis - torch.from_numpy(lvis.astype(np.float64))) ** 2).mean(-1)).mean()
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-2 * float(ref.detach()), (float(loss.detach()), float(ref.detach()))
    worst = max((_rel(got[k], P[k].grad.numpy()), k) for k in got)
    assert worst[0] < 5e-2, worst
    opt = optim.make_optimizer(model, cfg)
    losses = []
    for step in range(40):
        opt.zero_grad()
        pred, gt, kw, _ = model(batch, mode='train')
        l = model.compute_loss(pred, gt, **kw).mean()
        l.backward()
        losses.append(float(opt.step(loss=l.detach())))
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0], (losses[0], losses[-1])


@pytest.mark.parametrize("overrides,width,depth,skip,nf", [
    (dict(mlp_width='64', mlp_depth='3', mlp_skip_at='1'), 64, 3, 1, 2),
    (dict(n_freqs='4', z_dim='5'), 128, 4, 2, 4), (dict(pos_enc='False', mlp_width='96'), 96, 4, 2, 0)])
def test_brdf_plugin_non_shipped_shapes(nfx_lib, cuda, overrides, width, depth, skip, nf):
    """models.brdf.Model outside config/brdf.ini's architecture (reference brdf.py:57-86 builds width / depth / skip / bands
    from the ini, pos_enc = False = identity): both reciprocal halves against the oracle, the gradients of one training
    call — weights AND latent codes — against torch.autograd of the oracle, and AMSGrad reduces the loss."""
    from nerfactor_amd import optim
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from oracle import torch_train_ref as T
    cfg = make_config('brdf', lr='1e-3', **overrides)
    torch.manual_seed(5)
    model = get_model_class('brdf')(cfg)
    rng = np.random.default_rng(6)
    with torch.no_grad():
        for name, p in model.named_parameters():
            if name.endswith('.bias'):
                p.copy_(torch.from_numpy(rng.uniform(-.1, .1, size=tuple(p.shape)).astype(np.float32)))
    model = model.to(cuda)
    assert not model.tuned
    model.register_trainable()
    m = 700
    rusink = rng.uniform(0, np.pi / 2, size=(m, 3)).astype(np.float32)
    refl = rng.uniform(0.05, 2., size=(m, 1)).astype(np.float32)
    i = torch.zeros(m, dtype=torch.long, device=cuda)
    batch = (['a'] * m, i, None, None, None, dev(rusink, cuda), dev(refl, cuda))
    pred, gt, kw, _ = model(batch, mode='train')
    loss = model.compute_loss(pred, gt, **kw).mean()
    loss.backward()
    got = {k: p.grad.detach().cpu().numpy().copy() for k, p in model.named_parameters() if p.requires_grad}
    P = {k: p.detach().cpu().double().requires_grad_(True) for k, p in model.named_parameters()}
    zkey = [k for k in P if 'latent' in k or k.endswith('z') or 'code' in k]
    assert len(zkey) == 1, list(P)
    r64 = torch.from_numpy(rusink.astype(np.float64))
    z = P[zkey[0]][0:1].expand(m, -1)
    if cfg.getboolean('DEFAULT', 'normalize_z'):
        z = T.l2n(z, 1, 1e-12)
    T.QUANT = T.bf16_ste
    try:
        def run(r):
            h = T.mlp(torch.cat((z, T.embed(r, nf)), 1), P, 'brdf_mlp', depth, ['relu'] * depth, [skip])
            return T.mlp(h, P, 'brdf_out', 1, ['softplus'])
        brdf, reci = run(r64), run(torch.cat((r64[:, :1] + np.pi, r64[:, 1:]), 1))
    finally:
        T.QUANT = None
    assert np.abs(pred['brdf'].detach().cpu().numpy() - brdf.detach().numpy()).max() < 1e-2
    assert np.abs(pred['brdf_reci'].detach().cpu().numpy() - reci.detach().numpy()).max() < 1e-2
    f = {'log': torch.log, 'none': lambda v: v, 'divide': lambda v: v / (v + 1.)}[cfg.get('DEFAULT', 'loss_transform').lower()]
    g64 = torch.from_numpy(refl.astype(np.float64))
    ref = (((f(g64) - f(brdf)) ** 2).mean(-1) + ((f(g64) - f(reci)) ** 2).mean(-1)).mean()
    ref.backward()
    assert abs(float(loss.detach()) - float(ref.detach())) < 1e-2 * float(ref.detach())
    assert set(got) == set(P)
    worst = max((_rel(got[k], P[k].grad.numpy()), k) for k in got)
    assert worst[0] < 5e-2, worst
    opt = optim.make_optimizer(model, cfg)
    losses = []
    for step in range(40):
        opt.zero_grad()
        pred, gt, kw, _ = model(batch, mode='train')
        l = model.compute_loss(pred, gt, **kw).mean()
        l.backward()
        losses.append(float(opt.step(loss=l.detach())))
    assert np.isfinite(losses).all() and losses[-1] < 0.9 * losses[0], (losses[0], losses[-1])


@pytest.mark.parametrize("overrides,kw", [
    (dict(mlp_width='128', enc_depth='4', n_freqs_xyz='6'), dict(width=128, depth=4, n_freqs_xyz=6)),
    (dict(use_views='False', mlp_width='64'), dict(width=64, use_views=False, n_freqs_view=0)),
    (dict(mlp_width='128', enc_depth='4', n_freqs_xyz='6', precision='fp32'), dict(width=128, depth=4, n_freqs_xyz=6, fp32=True))])
def test_nerf_geometry_of_non_shipped_shapes(nfx_lib, cuda, overrides, kw):
    """Model.eval_sigma / eval_sigma_normal (what geometry_from_nerf.py:280-350 evaluates) for NeRFs outside the shipped
    architecture: density against the oracle's network, normals -l2_normalize(d relu(sigma)/dx) against torch.autograd
    in float64 with the kernels' bf16 operand rounding (straight-through) — the bounds of the tuned bf16 kernel's test
    (median cosine > 0.999, 10 % quantile > 0.98) — and loosely against plain float64."""
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from oracle import torch_train_ref as T
    cfg = make_config('nerf', **overrides)
    model = get_model_class('nerf')(cfg).to(cuda)
    assert not model.tuned
    rng = np.random.default_rng(21)
    lx, depth = kw.get('n_freqs_xyz', 10), kw.get('depth', 8)
    for pref in ('coarse_', 'fine_'):
        net = nerf_ref.init_nerf_net(rng, n_freqs_xyz=lx, n_freqs_view=kw.get('n_freqs_view', 4), width=kw.get('width', 256),
                                     depth=depth, sigma_bias=0.5, sigma_gain=8., use_views=kw.get('use_views', True))
        nerf_ref.randomize_biases(net, rng)
        for part, pairs in net.items():
            for layer, (k, b) in zip(model.net[pref + part].layers, pairs):
                with torch.no_grad():
                    layer.kernel.copy_(torch.from_numpy(k))
                    layer.bias.copy_(torch.from_numpy(b))
    n, s = 60, 11
    rayo = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    rayd = nerf_ref.l2_normalize(rng.normal(size=(n, 3)).astype(np.float32), 1, 1e-12)
    z = np.sort(rng.uniform(0.5, 3., size=(n, s)).astype(np.float32), 1)
    sig, normal = model.eval_sigma_normal(dev(rayo, cuda), dev(rayd, cuda), dev(z, cuda))
    sig_c = model.eval_sigma(dev(rayo, cuda), dev(rayd, cuda), dev(z, cuda), use_fine=True)
    assert torch.equal(sig, sig_c) and sig.shape == (n, s) and normal.shape == (n, s, 3)
    sig, normal = sig.cpu().numpy().reshape(-1), normal.cpu().numpy().reshape(-1, 3)
    pts = (rayo[:, None, :] + rayd[:, None, :] * z[:, :, None]).reshape(-1, 3)
    P = {k: p.detach().cpu().double() for k, p in model.named_parameters()}
    head = ('sigma_out', 1) if kw.get('use_views', True) else ('rgbs_out', 1)

    def reference(quant):
        T.QUANT = {True: T.bf16_ste, 'bf16': T.bf16_ste, 'pairs': T.pairs_ste, False: None, None: None}[quant]
        try:
            x = torch.tensor(pts, dtype=torch.float64, requires_grad=True)
            feat = T.mlp(T.embed(x, lx), P, 'fine_enc', depth, ['relu'] * depth, [depth // 2])
            raw = T.mlp(feat, P, 'fine_' + head[0], 1, [None])[:, -1]
        finally:
            T.QUANT = None
        (g,) = torch.autograd.grad(torch.relu(raw).sum(), x)
        return raw.detach().numpy(), g.numpy()
    raw_q, _ = reference(not kw.get('fp32'))
    assert np.abs(np.maximum(raw_q, 0) - sig).max() < (1e-4 if kw.get('fp32') else 2e-2) * max(1., np.abs(raw_q).max())
    on = sig > 0
    assert 0.2 < on.mean() < 1.0
    assert np.abs(normal[~on]).max() == 0.
    np.testing.assert_allclose(np.linalg.norm(normal[on], axis=1), 1., atol=1e-5)
    # (precision = fp32: the fp32 instantiation against plain float64 — six nines)
    for quant, (p50, p10) in (((False, (0.999999, 0.99999)),) if kw.get('fp32') else ((True, (0.999, 0.98)), (False, (0.99, 0.8)))):
        raw, g = reference(quant)
        want = -g / np.maximum(np.linalg.norm(g, axis=1, keepdims=True), 1e-30)
        both = on & (raw > 0)
        cos = (want[both] * normal[both]).sum(1)
        assert np.median(cos) > p50 and np.quantile(cos, 0.1) > p10, (quant, np.median(cos), np.quantile(cos, 0.1))


def test_embed_backward_and_input_gradient_only_mode(nfx_lib, cuda):
    """nfx_embed_bwd against the analytic pull-back; nfx_mlp_generic_bwd without gradient buffers returns the same dx as
    with them and touches nothing else."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(2)
    x = rng.uniform(-2, 2, size=(300, 3))
    for L in (0, 3, 10):
        d = rng.normal(size=(300, 3 + 6 * L + 5))
        got = ops.embed_bwd(L, dev(x, cuda), dev(d, cuda), col0=2).cpu().numpy()
        want = d[:, 2:5].copy()
        for k in range(L):
            c = 5 + 6 * k
            want += 2. ** k * (np.cos(2. ** k * x) * d[:, c:c + 3] - np.sin(2. ** k * x) * d[:, c + 3:c + 6])
        # (fp32 arguments: the 2^9 band alone turns the rounding of x into 1e-4 rad, weighted 2^9 again)
        assert np.abs(got - want).max() < 1e-4 * np.abs(want).max(), (L, np.abs(got - want).max(), np.abs(want).max())
    ks = [nerf_ref.glorot_uniform(rng, 20, 48), nerf_ref.glorot_uniform(rng, 48, 2)]
    bs = [np.zeros(48, np.float32), np.zeros(2, np.float32)]
    net = ops.GenericNet(ks, bs, ['relu', None], train=True).to(cuda)
    xin, dy = dev(rng.normal(size=(500, 20)), cuda), dev(rng.normal(size=(500, 2)), cuda)
    dks, dbs = [torch.zeros(k.shape, device=cuda) for k in ks], [torch.zeros(b.shape, device=cuda) for b in bs]
    a = ops.mlp_generic_bwd(xin, net, dy, dks, dbs, want_dx=True)
    b = ops.mlp_generic_bwd(xin, net, dy, None, None, want_dx=True)
    assert torch.equal(a, b)
    with pytest.raises(Exception, match='nothing to compute'):
        ops.mlp_generic_bwd(xin, net, dy, None, None)


def test_nerf_plugin_non_shipped_shape_at_fp32(nfx_lib, cuda):
    """precision = fp32 of a non-shipped NeRF (128 x 4, 6 / 2 bands): the fp32 instantiation of the runtime-shaped kernels
    renders it within the fp32 tolerance of SURVEY.md section 8d (2e-4 on rgb) on every ray off the alpha_last discontinuity."""
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    cfg = make_config('nerf', mlp_width='128', enc_depth='4', n_freqs_xyz='6', n_freqs_view='2', precision='fp32')
    model = get_model_class('nerf')(cfg).to(cuda)
    assert not model.tuned and model.grad_precision == 'fp32'
    rng = np.random.default_rng(5)
    nets = []
    for pref in ('coarse_', 'fine_'):
        net = nerf_ref.init_nerf_net(rng, n_freqs_xyz=6, n_freqs_view=2, width=128, depth=4, sigma_bias=0.5, sigma_gain=8.)
        nerf_ref.randomize_biases(net, rng)
        nets.append(net)
        for part, pairs in net.items():
            for layer, (k, b) in zip(model.net[pref + part].layers, pairs):
                with torch.no_grad():
                    layer.kernel.copy_(torch.from_numpy(k))
                    layer.bias.copy_(torch.from_numpy(b))
    rayo, rayd = common.camera_rays(12, 12)
    n = rayo.shape[0]
    batch = (['v'] * n, torch.tensor([[12, 12]] * n), dev(rayo, cuda), dev(rayd, cuda), torch.rand(n, 3, device=cuda))
    pred, _, _, _ = model(batch, mode='test')
    coarse, fine, aux = nerf_ref.render_rays(rayo, rayd, nets[0], nets[1], n_freqs_xyz=6, n_freqs_view=2)
    ok = (np.abs(aux['rgbs_coarse'][:, -1, 3]) > 1e-3) & (np.abs(aux['rgbs_fine'][:, -1, 3]) > 1e-3)
    assert ok.mean() > 0.9
    err_c = np.abs(pred['coarse'].cpu().numpy() - coarse['rgb']).max(-1)
    assert err_c[ok].max() <= 2e-4, err_c[ok].max()
    # (the fine pass adds the inverse-CDF bin edges as a second discontinuity: bounded on the bulk of the rays)
    err_f = np.abs(pred['fine'].cpu().numpy() - fine['rgb']).max(-1)
    assert np.quantile(err_f[ok], 0.9) <= 2e-4 and err_f[ok].max() <= 3e-2, (np.quantile(err_f[ok], 0.9), err_f[ok].max())


def test_accumulate_sigma_with_another_far_distance(nfx_lib, cuda):
    """Model.accumulate_sigma(sigma, z, rayd, inf=...) (nerf.py:184-212): `inf` is the distance the LAST sample is given.  The
    compositing kernel has 1e10 built in; any other value goes through the same kernel on S + 1 samples (an empty sample
    `inf` behind the last one).  Against the reference formula in float64."""
    from nerfactor_amd.nerfactor.models import get_model_class
    rng = np.random.default_rng(0)
    n, s = 300, 17
    sigma = rng.normal(size=(n, s)).astype(np.float32) * 3
    z = np.sort(rng.uniform(2, 6, size=(n, s)).astype(np.float32), 1)
    rayd = rng.normal(size=(n, 3)).astype(np.float32)
    Model = get_model_class('nerf')
    for inf in (1e10, 0.25, 7.5, 1e3):
        got = Model.accumulate_sigma(dev(sigma, cuda), dev(z, cuda), dev(rayd, cuda), inf=inf).cpu().numpy()
        dist = np.concatenate([np.diff(z.astype(np.float64), axis=1), np.full((n, 1), inf)], 1) * np.linalg.norm(rayd.astype(np.float64), axis=1, keepdims=True)
        alpha = 1. - np.exp(-np.maximum(sigma.astype(np.float64), 0) * dist)
        trans = np.cumprod(np.concatenate([np.ones((n, 1)), (1. - alpha + 1e-6)[:, :-1]], 1), 1)
        want = alpha * trans
        assert got.shape == (n, s) and np.abs(got - want).max() < 2e-5, (inf, np.abs(got - want).max())


def test_destination_and_input_shapes_are_validated(nfx_lib, cuda):
    """`out` / `col0` destinations and input widths that do not fit are refused before a launch (they would write or read
    past a row: ops._check_out, autograd.GenericMlp.forward)."""
    from nerfactor_amd import autograd, ops
    rng = np.random.default_rng(4)
    ks = [nerf_ref.glorot_uniform(rng, 20, 48), nerf_ref.glorot_uniform(rng, 48, 2)]
    bs = [np.zeros(48, np.float32), np.zeros(2, np.float32)]
    net = ops.GenericNet(ks, bs, ['relu', None], train=True).to(cuda)
    x = dev(rng.normal(size=(64, 20)), cuda)
    with pytest.raises(Exception, match='do not fit'):
        ops.mlp_generic_fwd(x, net, out=torch.zeros(64, 5, device=cuda), col0=4)
    with pytest.raises(Exception, match='out must be'):
        ops.mlp_generic_fwd(x, net, out=torch.zeros(63, 5, device=cuda))
    with pytest.raises(Exception, match='out must be'):
        ops.mlp_generic_fwd(x, net, out=torch.zeros(64, 5))
    with pytest.raises(Exception, match='do not fit'):
        ops.embed(2, x=x[:, :3].contiguous(), out=torch.zeros(64, 15, device=cuda), col0=1)
    with pytest.raises(Exception, match='out must be'):
        ops.embed(2, x=x[:, :3].contiguous(), out=torch.zeros(64, 15, device=cuda, dtype=torch.float64))
    got = ops.embed(2, x=x[:, :3].contiguous(), out=torch.zeros(64, 17, device=cuda), col0=2)
    assert torch.equal(got[:, 2:], ops.embed(2, x=x[:, :3].contiguous())) and not got[:, :2].any()
    dy = dev(rng.normal(size=(64, 2)), cuda)
    with pytest.raises(Exception, match='the network reads'):
        ops.mlp_generic_bwd(x[:, :19], net, dy, None, None, want_dx=True)
    wide = dev(rng.normal(size=(64, 24)), cuda).requires_grad_()
    with pytest.raises(Exception, match="the network's input width"):
        autograd.GenericMlp.apply(wide, lambda: net)
    assert torch.equal(ops.mlp_generic_fwd(wide.detach(), net), ops.mlp_generic_fwd(wide.detach()[:, :20].contiguous(), net))
