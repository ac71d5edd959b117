"""Differentiable torch-CPU restatement of the reference's TRAINING losses (TEST INFRASTRUCTURE, see
oracle/__init__.py).  torch.autograd of these functions is the oracle for row a18 (train step); it is pinned to the
reference's own training step differentiated by tests/golden/make_reference_grad_golden.py
(tests/golden/reference_grads.npz, tests/test_cpu_reference_grads.py).

  nerf_loss        models/nerf.py:99-118,149-182,184-254,292-300 (train mode: stratified coarse samples, sigma noise,
                   stochastic inverse-CDF fine samples behind tf.stop_gradient), util/math.py:63-94
  nerfactor_loss   models/nerfactor.py:181-541 (train mode with the jittered second evaluation and every smoothness /
                   light prior), models/shape.py:128-237, nerfactor_microfacet.py:108-124, microfacet.py:30-111,
                   util/geom.py:119-192 with the custom gradients of util/math.py:24-60
  KerasAMSGrad     tf.keras.optimizers.Adam(amsgrad=True) + ExponentialDecay as trainvali.py:110-127 builds them
Parameters are dicts {state_dict key: tensor}; dtype follows the parameters (float64 anchor / float32 mirror).
"""
import math

import torch


def embed(x, n_freqs):
    parts = [x]
    for k in range(n_freqs):
        parts += [torch.sin(x * 2. ** k), torch.cos(x * 2. ** k)]
    return torch.cat(parts, -1)


def l2n(x, dim, eps):
    return x * torch.rsqrt(tf_maximum((x * x).sum(dim, keepdim=True), eps))


def tf_maximum(a, b):
    """tf.maximum's gradient: everything to `a` where a >= b."""
    b = torch.as_tensor(b, dtype=a.dtype)
    return torch.where(a >= b, a, b.expand_as(a))


def divide_no_nan(a, b):
    a, b = torch.broadcast_tensors(a, b)
    zero = b == 0
    return torch.where(zero, torch.zeros_like(a), a / torch.where(zero, torch.ones_like(b), b))


_ACT = {None: lambda v: v, 'relu': torch.relu, 'sigmoid': torch.sigmoid, 'softplus': torch.nn.functional.softplus}

# Optional operand rounding of every Dense layer (weights and layer inputs to bf16, fp32/fp64 accumulate) with a
# straight-through gradient: what the MFMA path computes.  Lets a test separate "the kernels differ from the reference"
# from "a bf16 forward differs from an fp32 forward" (ReLU masks and the sign of L1 smoothness terms flip).
QUANT = None


def bf16_ste(t):
    return t + (t.detach().float().to(torch.bfloat16).to(t.dtype) - t.detach())


def pairs_ste(t):
    """The fp32-class operand of csrc/mlp_x3.hpp / mlp_generic.hip (NFX_PREC_FP32): the fp32 value as a bf16 hi / lo pair —
    hi = bf16(v), lo = bf16(v - hi), 16 significant bits — with a straight-through gradient."""
    v = t.detach().float()
    hi = v.to(torch.bfloat16).float()
    lo = (v - hi).to(torch.bfloat16).float()
    return t + ((hi + lo).to(t.dtype) - t.detach())


def mlp(x, P, name, n_layers, acts, skip_at=None):
    q = QUANT if QUANT is not None else (lambda t: t)
    h = x
    for i in range(n_layers):
        h = _ACT[acts[i]](q(h) @ q(P['net_%s_layer%d.kernel' % (name, i)]) + P['net_%s_layer%d.bias' % (name, i)])
        if skip_at is not None and i in skip_at:
            h = torch.cat((h, x), -1)
    return h


def mlp128(x, P, name, out_act):
    h = mlp(x, P, name + '_mlp', 4, ['relu'] * 4, skip_at=[2])
    return mlp(h, P, name + '_out', 1, [out_act])


# ------------------------------------------------------------------------------------------------ NeRF
def _nerf_net(pts, views, P, pref, arch=None):
    """arch = (n_freqs_xyz, n_freqs_view, enc_depth, use_views), default config/nerf.ini's; the widths are the
    parameters' (models/nerf.py:53-90)."""
    lx, lv, depth, use_views = arch or (10, 4, 8, True)
    pe, ve = embed(pts.reshape(-1, 3), lx), embed(views.reshape(-1, 3), lv)
    feat = mlp(pe, P, pref + 'enc', depth, ['relu'] * depth, skip_at=[depth // 2])
    if not use_views:
        return mlp(feat, P, pref + 'rgbs_out', 1, [None]).reshape(pts.shape[:2] + (4,))
    sigma = mlp(feat, P, pref + 'sigma_out', 1, [None])
    feat = mlp(feat, P, pref + 'bottleneck', 1, [None])
    rgb = mlp(torch.cat((feat, ve), -1), P, pref + 'rgb_out', 2, ['relu', None])
    return torch.cat((rgb, sigma), -1).reshape(pts.shape[:2] + (4,))


def _accumulate(rgbs, z, rayd, noise, white_bg):
    dist = z[:, 1:] - z[:, :-1]
    dist = torch.cat((dist, torch.full_like(dist[:, :1], 1e10)), -1)
    dist = dist * torch.sqrt((rayd[:, None, :] ** 2).sum(-1))
    density = 1. - torch.exp(-torch.relu(rgbs[:, :, 3] + noise) * dist)
    x = 1. - density + 1e-6
    excl = torch.cat((torch.ones_like(x[:, :1]), torch.cumprod(x, -1)[:, :-1]), -1)
    weights = density * excl
    rgb = torch.sigmoid(rgbs[:, :, :3])
    occu = weights.sum(-1)
    rgb = (weights[:, :, None] * rgb).sum(-2)
    bg = 1. if white_bg else 0.
    return rgb * occu[:, None] + bg * (1. - occu[:, None]), weights


def nerf_loss(P, rayo, rayd, gt, u_coarse, n_coarse_noise, u_fine, n_fine_noise, near=2., far=6., n_coarse=64,
              n_fine=128, white_bg=True, noise_std=0., arch=None):
    """Per-ray training loss of models/nerf.py (loss = l2, keep_batch): the draws are the tf.random tensors in the order
    the reference makes them (uniform [n, 64], normal [n, 64], uniform [n, 128], normal [n, 192])."""
    dt = rayo.dtype
    rayd = rayd * torch.rsqrt(tf_maximum((rayd * rayd).sum(1, keepdim=True), 1e-12))
    n = rayo.shape[0]
    t = (torch.arange(n_coarse, dtype=torch.float32) * (1. / (n_coarse - 1))).to(dt)
    t[-1] = 1.
    z = (near * (1. - t) + far * t).expand(n, n_coarse)
    mid = .5 * (z[:, 1:] + z[:, :-1])
    upper, lower = torch.cat((mid, z[:, -1:]), -1), torch.cat((z[:, :1], mid), -1)
    z = lower + (upper - lower) * u_coarse
    pts = rayo[:, None, :] + rayd[:, None, :] * z[:, :, None]
    rgbs = _nerf_net(pts, rayd[:, None, :].expand(pts.shape), P, 'coarse_', arch)
    rgb_c, w = _accumulate(rgbs, z, rayd, n_coarse_noise * noise_std, white_bg)
    # gen_z_fine: inverse-transform sampling of the coarse weights, no gradient (nerf.py:143)
    with torch.no_grad():
        mid = .5 * (z[:, 1:] + z[:, :-1])
        wts = w[:, 1:-1]
        pdf = wts / (wts.sum(-1, keepdim=True) + 1e-5)
        cdf = torch.cat((torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, -1)), -1)
        ind = torch.searchsorted(cdf.contiguous(), u_fine.contiguous(), right=True)
        below, above = torch.clamp(ind - 1, min=0), torch.clamp(ind, max=cdf.shape[-1] - 1)
        cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
        val_b, val_a = torch.gather(mid, 1, below), torch.gather(mid, 1, above)
        den = cdf_a - cdf_b
        den = torch.where(den < 1e-5, torch.ones_like(den), den)
        z_fine = val_b + (u_fine - cdf_b) / den * (val_a - val_b)
    z_all = torch.sort(torch.cat((z, z_fine), -1), -1).values
    pts = rayo[:, None, :] + rayd[:, None, :] * z_all[:, :, None]
    rgbs = _nerf_net(pts, rayd[:, None, :].expand(pts.shape), P, 'fine_', arch)
    rgb_f, _ = _accumulate(rgbs, z_all, rayd, n_fine_noise * noise_std, white_bg)
    return ((gt - rgb_c) ** 2).mean(-1) + ((gt - rgb_f) ** 2).mean(-1)


# ------------------------------------------------------------------------------------------------ NeRFactor
class SafeAcos(torch.autograd.Function):      # util/math.py:41-60
    @staticmethod
    def forward(ctx, x):
        xc = torch.clamp(x, -1., 1.)
        ctx.save_for_backward(xc)
        return torch.acos(xc)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        return dy * (-1. / (torch.sqrt(1. - xc ** 2 + 1e-6) + 1e-6))


class SafeAtan2(torch.autograd.Function):     # util/math.py:24-38
    @staticmethod
    def forward(ctx, x, y):
        ctx.save_for_backward(x, y)
        return torch.atan2(x, y)

    @staticmethod
    def backward(ctx, dz):
        x, y = ctx.saved_tensors
        den = x ** 2 + y ** 2 + 1e-6
        return dz * y / den, dz * (-x / den)


def _microfacet(l, v, n, albedo, rough, f0):
    l, v, n = l2n(l, 2, 1e-6), l2n(v, 1, 1e-6), l2n(n, 1, 1e-6)
    h = l2n(l + v[:, None, :], 2, 1e-6)
    f = f0 + (1 - f0) * (1 - (l * h).sum(-1)) ** 5
    alpha = rough ** 2
    cos_m = torch.einsum('ijk,ik->ij', h, n)
    chi = (cos_m > 0).to(l.dtype)
    cos_m_sq = cos_m ** 2
    tan_m_sq = divide_no_nan(1 - cos_m_sq, cos_m_sq)
    d = divide_no_nan(alpha ** 2 * chi, math.pi * cos_m_sq ** 2 * (alpha ** 2 + tan_m_sq) ** 2)
    cos_v = (n * v).sum(-1)
    cos_t = torch.einsum('ijk,ik->ij', h, v)
    chi_g = (divide_no_nan(cos_t, cos_v[:, None].expand_as(cos_t)) > 0).to(l.dtype)
    cos_v_sq = torch.clamp(cos_v ** 2, 0., 1.)
    tan_v_sq = torch.clamp(divide_no_nan(1 - cos_v_sq, cos_v_sq), min=0.)
    g = divide_no_nan(chi_g * 2, 1 + torch.sqrt(1 + alpha ** 2 * tan_v_sq[:, None]))
    l_dot_n = torch.einsum('ijk,ik->ij', l, n)
    spec = divide_no_nan(f * g * d, 4 * l_dot_n.abs() * cos_v.abs()[:, None])
    return albedo[:, None, :] / math.pi + spec[:, :, None]


def _rot_vec(v, axis, ang):
    axis = torch.tensor(axis, dtype=v.dtype).reshape(1, 3)
    c, s = torch.cos(ang)[:, None], torch.sin(ang)[:, None]
    return v * c + axis * (v @ axis.T) * (1 - c) + torch.cross(axis.expand_as(v), v, dim=1) * s


def _learned_brdf(l, v, n, albedo, z, PB, scale):
    nn = l2n(n, 1, 1e-6)
    up = (torch.tensor((0., 0., 1.), dtype=n.dtype) + 1e-6).expand_as(nn)
    t = l2n(torch.cross(nn, up, dim=1), 1, 1e-6)
    b = l2n(torch.cross(nn, t, dim=1), 1, 1e-6)
    rot = torch.stack((t, b, nn), 1)
    vdir = torch.einsum('jkl,jl->jk', rot, v)
    ldir = torch.einsum('jkl,jnl->jnk', rot, l)
    n_pts, nl = ldir.shape[:2]
    lf = ldir.reshape(-1, 3)
    vf = vdir[:, None, :].expand(n_pts, nl, 3).reshape(-1, 3)
    a, bb = l2n(lf, 1, 1e-6), l2n(vf, 1, 1e-6)
    h = l2n((a + bb) / 2, 1, 1e-6)
    theta_h = SafeAcos.apply(h[:, 2])
    phi_h = SafeAtan2.apply(h[:, 1], h[:, 0])
    diff = _rot_vec(_rot_vec(bb, (0., 0., 1.), -phi_h), (0., 1., 0.), -theta_h)
    theta_d = SafeAcos.apply(diff[:, 2])
    phi_d = torch.remainder(SafeAtan2.apply(diff[:, 1], diff[:, 0]), math.pi)
    rus = torch.stack((phi_d, theta_h, theta_d), 1)
    zf = z[:, None, :].expand(n_pts, nl, z.shape[1]).reshape(-1, z.shape[1])
    front = lf[:, 2] > 0
    x = torch.cat((zf, embed(rus, 2)), 1)[front]
    y = mlp128(x, PB, 'brdf', 'softplus')
    spec = torch.zeros(n_pts * nl, 1, dtype=l.dtype).index_put((torch.nonzero(front)[:, 0],), y)
    return albedo[:, None, :] / math.pi + spec.reshape(n_pts, nl, 1).expand(n_pts, nl, 3) * scale


def linear2srgb(x):
    x = torch.clamp(x, 0., 1.)
    return torch.where(x <= 0.0031308, x * 12.92, 1.055 * torch.pow(x, 1 / 2.4) - 0.055)


def nerfactor_loss(P, batch, xyz_noise, lxyz, lareas, hp, variant='microfacet', PB=None):
    """Per-ray training loss of models/nerfactor{,_microfacet}.py.  batch = (rayo, rgb, alpha, xyz, normal, lvis);
    hp = the ini hyper-parameters (normal_loss_weight, lvis_loss_weight, normal_smooth_weight, lvis_smooth_weight,
    albedo_smooth_weight, brdf_smooth_weight, albedo_slope, albedo_bias, light_tv_weight, light_achro_weight, fresnel_f0
    | learned_brdf_scale, white_bg, linear2srgb, smooth_use_l1); PB = the frozen BRDF prior's parameters."""
    rayo, rgb, alpha, xyz, normal, lvis = batch
    dt = xyz.dtype
    mask = alpha[:, 0] > 0
    xm, cm = xyz[mask], rayo[mask]
    surf2l = l2n(lxyz.reshape(1, -1, 3) - xm[:, None, :], 2, 1e-6)
    surf2c = l2n(cm - xm, 1, 1e-6)
    nl = surf2l.shape[1]
    z_act = 'sigmoid' if variant == 'microfacet' else None

    def heads(p):
        pe = embed(p, 10)
        nrm = l2n(mlp128(pe, P, 'normal', None) + 1e-6, 1, 1e-6)
        rows = torch.cat((embed(p[:, None, :].expand(-1, nl, -1).reshape(-1, 3), 10), embed(surf2l.reshape(-1, 3), 4)), -1)
        lv = mlp128(rows, P, 'lvis', 'sigmoid').reshape(-1, nl)
        alb = hp['albedo_slope'] * mlp128(pe, P, 'albedo', 'sigmoid') + hp['albedo_bias']
        return nrm, lv, alb, mlp128(pe, P, 'brdf_z', z_act)

    nrm, lv, alb, z = heads(xm)
    nrm_j, lv_j, alb_j, z_j = heads(xm + xyz_noise)
    light = torch.clamp(P['_light'], min=0.)
    if variant == 'microfacet':
        brdf = _microfacet(surf2l, surf2c, nrm, alb, z, hp['fresnel_f0'])
    else:
        brdf = _learned_brdf(surf2l, surf2c, nrm, alb, z, PB, hp['learned_brdf_scale'])
    cos = torch.einsum('ijk,ik->ij', surf2l, nrm)
    lvm = (cos > 0).to(dt) * lv
    contrib = brdf * (lvm[:, :, None] * light.reshape(1, -1, 3)) * cos[:, :, None] * lareas.reshape(1, -1, 1)
    rgb_pred = torch.clamp(contrib.sum(1), 0., 1.)
    if hp['linear2srgb']:
        rgb_pred = linear2srgb(rgb_pred)
    n_all = alpha.shape[0]
    idx = torch.nonzero(mask)[:, 0]

    def full(v):
        return torch.zeros((n_all,) + tuple(v.shape[1:]), dtype=dt).index_put((idx,), v)

    bg = 1. if hp['white_bg'] else 0.
    on_bg = lambda x: x * alpha + bg * (1. - alpha)
    mse = lambda a, b: ((a - b) ** 2).mean(-1)
    smooth = (lambda a, b: (a - b).abs().mean(-1)) if hp['smooth_use_l1'] else mse
    rgb_p, rgb_g = on_bg(full(rgb_pred)), on_bg(full(rgb[mask]))
    n_p, n_g = on_bg(full(nrm)), on_bg(full(normal[mask]))
    v_p, v_g = on_bg(full(lv)), on_bg(full(lvis[mask]))
    loss = mse(rgb_g, rgb_p) + hp['normal_loss_weight'] * mse(n_g, n_p) + hp['lvis_loss_weight'] * mse(v_g, v_p)
    loss = loss + hp['normal_smooth_weight'] * smooth(n_p, full(nrm_j)) + hp['lvis_smooth_weight'] * smooth(v_p, full(lv_j))
    loss = loss + hp['albedo_smooth_weight'] * smooth(full(alb), full(alb_j))
    loss = loss + hp['brdf_smooth_weight'] * smooth(full(z), full(z_j))
    if hp['light_tv_weight'] > 0:
        dx, dy = light - torch.roll(light, 1, 1), light - torch.roll(light, 1, 0)
        loss = loss + hp['light_tv_weight'] * (dx ** 2 + dy ** 2).sum()
    if hp['light_achro_weight'] > 0:
        dc = light - torch.roll(light, 1, 2)
        loss = loss + hp['light_achro_weight'] * (dc ** 2).sum()
    return loss


# ------------------------------------------------------------------------------------------------ optimizer
def brdf_prior_loss(P, ind, rusink, refl):
    """models/brdf.py:87-136 in train mode with loss = l2 on log reflectance (config/brdf.ini), keep_batch=True
    (trainvali.py:280): per row (log refl - log brdf)^2 + (log refl - log brdf_reci)^2, brdf_reci at phi_d + pi.
    P holds net_brdf_{mlp,out}_layer*.{kernel,bias} and the latent table 'latent_code._z' [n_brdfs, z_dim]."""
    z = P['latent_code._z'][ind.long()]          # tf.gather_nd of the variable: scatter-add gradient
    def run(r):
        return mlp128(torch.cat((z, embed(r, 2)), 1), P, 'brdf', 'softplus')
    reci = torch.cat((rusink[:, :1] + math.pi, rusink[:, 1:]), 1)
    brdf, brdf_reci = run(rusink), run(reci)
    lg = torch.log(refl)
    return ((lg - torch.log(brdf)) ** 2).mean(1) + ((lg - torch.log(brdf_reci)) ** 2).mean(1)


class KerasAMSGrad:
    """tf.keras.optimizers.Adam(learning_rate = lr | ExponentialDecay(lr, decay_steps, decay_rate), amsgrad=True) on a
    dict of tensors: t = iterations + 1, lr_t = lr(iterations) sqrt(1 - b2^t) / (1 - b1^t), epsilon 1e-7 outside the
    square root (TF 2.2 optimizer_v2/adam.py)."""
    def __init__(self, lr, decay_steps=-1, decay_rate=1., beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self.lr, self.decay_steps, self.decay_rate = lr, decay_steps, decay_rate
        self.b1, self.b2, self.eps, self.iterations, self.state = beta_1, beta_2, epsilon, 0, {}

    def step(self, params, grads):
        lr = self.lr * self.decay_rate ** (self.iterations / self.decay_steps) if self.decay_steps > 0 else self.lr
        t = self.iterations + 1
        lr_t = lr * math.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
        with torch.no_grad():
            for k, p in params.items():
                g = grads[k]
                m, v, vhat = self.state.setdefault(k, [torch.zeros_like(p) for _ in range(3)])
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                torch.maximum(vhat, v, out=vhat)
                p.sub_(lr_t * m / (vhat.sqrt() + self.eps))
        self.iterations += 1
