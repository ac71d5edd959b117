"""torch-CPU fp32 restatement of the hot path (TEST INFRASTRUCTURE / the timed CPU baseline; see
oracle/__init__.py — parity with TensorFlow itself is UNPINNED).

This is the stand-in for the reference's own `--device cpu` path (trainvali.py:263-264): same op
sequence and the same Python-level chunking (`mlp_chunk = accu_chunk = 65536`,
config/nerf.ini:65-66), dense layers through the multi-threaded CPU BLAS exactly as TF-CPU would
run them through Eigen/MKL.  Second, independent restatement next to oracle/nerf_ref.py (NumPy);
tests/ cross-check the two.

  nerfactor/models/nerf.py:120-290, nerfactor/util/math.py:63-94,
  nerfactor/networks/embedder.py:23-47, nerfactor/networks/mlp.py:24-50
"""
import torch


def to_torch_net(net):
    return {k: [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in v] for k, v in net.items()}


def embed(x, n_freqs):
    parts = [x]
    for k in range(n_freqs):
        f = float(2 ** k)
        parts += [torch.sin(x * f), torch.cos(x * f)]
    return torch.cat(parts, -1)


def mlp(x, layers, acts, skip_at=None):
    h = x
    for i, (w, b) in enumerate(layers):
        h = h @ w + b
        if acts[i] == 'relu':
            h = torch.relu(h)
        elif acts[i] == 'sigmoid':
            h = torch.sigmoid(h)
        elif acts[i] == 'softplus':
            h = torch.nn.functional.softplus(h)
        if skip_at is not None and i in skip_at:
            h = torch.cat((h, x), -1)
    return h


def eval_nerf_at(pts, views, net, mlp_chunk=65536):
    depth = len(net['enc'])
    pf, vf = pts.reshape(-1, 3), views.reshape(-1, 3)
    out = []
    for i in range(0, pf.shape[0], mlp_chunk):
        pe = embed(pf[i:i + mlp_chunk], 10)
        ve = embed(vf[i:i + mlp_chunk], 4)
        feat = mlp(pe, net['enc'], ['relu'] * depth, skip_at=[depth // 2])
        sigma = mlp(feat, net['sigma_out'], [None])
        feat = mlp(feat, net['bottleneck'], [None])
        rgb = mlp(torch.cat((feat, ve), -1), net['rgb_out'], ['relu', None])
        out.append(torch.cat((rgb, sigma), -1))
    return torch.cat(out, 0).reshape(pts.shape[:2] + (4,))


def gen_z(near, far, n_samples, n_rays):
    t = torch.arange(n_samples, dtype=torch.float32) * (1. / (n_samples - 1))
    z = near * (1. - t) + far * t
    return z.expand(n_rays, n_samples).contiguous()


def accumulate(rgbs, z, rayd, white_bg=True, accu_chunk=65536):
    dist = z[:, 1:] - z[:, :-1]
    dist = torch.cat((dist, torch.full_like(dist[:, :1], 1e10)), -1)
    dist = dist * torch.linalg.norm(rayd[:, None, :], dim=-1)
    density = 1. - torch.exp(-torch.relu(rgbs[:, :, 3]) * dist)
    ws = []
    for i in range(0, density.shape[0], accu_chunk):
        d = density[i:i + accu_chunk]
        x = 1. - d + 1e-6
        excl = torch.cat((torch.ones_like(x[:, :1]), torch.cumprod(x, -1)[:, :-1]), -1)
        ws.append(d * excl)
    weights = torch.cat(ws, 0)
    rgb = torch.sigmoid(rgbs[:, :, :3])
    occu = weights.sum(-1)
    rgb = (weights[:, :, None] * rgb).sum(-2)
    depth = (weights * z).sum(-1)
    disp = 1. / torch.clamp(depth, min=1e-10)
    bg = torch.ones_like(rgb) if white_bg else torch.zeros_like(rgb)
    rgb = rgb * occu[:, None] + bg * (1. - occu[:, None])
    return rgb, occu, depth, disp, weights


def inv_transform_sample(val, weights, n_samples, eps=1e-5):
    denom = weights.sum(-1, keepdim=True) + eps
    pdf = weights / denom
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat((torch.zeros_like(cdf[:, :1]), cdf), -1)
    u = (torch.arange(n_samples, dtype=torch.float32) * (1. / (n_samples - 1)))
    u = u.expand(cdf.shape[0], n_samples).contiguous()
    ind = torch.searchsorted(cdf.contiguous(), u, right=True)
    below = torch.clamp(ind - 1, min=0)
    above = torch.clamp(ind, max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    val_b, val_a = torch.gather(val, 1, below), torch.gather(val, 1, above)
    den = cdf_a - cdf_b
    den = torch.where(den < eps, torch.ones_like(den), den)
    t = (u - cdf_b) / den
    return val_b + t * (val_a - val_b)


def render_rays(rayo, rayd, net_coarse, net_fine, near=2., far=6., n_coarse=64, n_fine=128,
                white_bg=True):
    """Model._render_rays (nerf.py:149-182), perturb=False.  All torch-CPU fp32."""
    rayd = rayd * torch.rsqrt(torch.clamp((rayd * rayd).sum(1, keepdim=True), min=1e-12))
    z = gen_z(near, far, n_coarse, rayo.shape[0])
    pts = rayo[:, None, :] + rayd[:, None, :] * z[:, :, None]
    views = rayd[:, None, :].expand(pts.shape)
    rgbs = eval_nerf_at(pts, views, net_coarse)
    rgb_c, occu_c, depth_c, _, w = accumulate(rgbs, z, rayd, white_bg)
    sig_c = rgbs[:, -1, 3].abs()
    mid = .5 * (z[:, 1:] + z[:, :-1])
    z_fine = inv_transform_sample(mid, w[:, 1:-1], n_fine)
    z_all, _ = torch.sort(torch.cat((z, z_fine), -1), -1)
    pts = rayo[:, None, :] + rayd[:, None, :] * z_all[:, :, None]
    views = rayd[:, None, :].expand(pts.shape)
    rgbs = eval_nerf_at(pts, views, net_fine)
    rgb_f, occu_f, depth_f, _, _ = accumulate(rgbs, z_all, rayd, white_bg)
    return {'rgb': rgb_c, 'occu': occu_c, 'depth': depth_c}, \
           {'rgb': rgb_f, 'occu': occu_f, 'depth': depth_f}, \
           {'z_all': z_all, 'sigma_last_coarse': sig_c, 'sigma_last_fine': rgbs[:, -1, 3].abs()}


# ------------------------------------------------------------------------------------------------------------
# NeRFactor surface-shading stage (models/shape.py:128-237, models/nerfactor.py:181-461, nerfactor_microfacet.py,
# brdf/microfacet/microfacet.py:30-111, util/geom.py:119-192, util/img.py:140-163) in torch-CPU fp32: the timed
# CPU baseline of bench.py's NeRFactor leg.  Same op sequence and chunking as the reference (chunk_apply over
# mlp_chunk rows of the flattened N*L (point, light) table); checked against oracle/nerfactor_ref.py in
# tests/test_cpu_nerfactor.py.
def _safe_l2n(x, dim, eps=1e-6):
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim, keepdim=True), min=eps))


def _chunk_apply(func, x, chunk):
    return torch.cat([func(x[i:i + chunk]) for i in range(0, x.shape[0], chunk)], 0)


def _mlp128(x, net, name, out_act):
    h = mlp(x, net[name + '_mlp'], ['relu'] * 4, skip_at=[2])
    return mlp(h, net[name + '_out'], [out_act])


def _divide_no_nan(a, b):
    return torch.where(b == 0, torch.zeros_like(a * b), a / torch.where(b == 0, torch.ones_like(b), b))


def _microfacet(pts2l, pts2c, normal, albedo, rough, f0):
    h = _safe_l2n(pts2l + pts2c[:, None, :], 2)
    f = f0 + (1 - f0) * (1 - (pts2l * h).sum(-1)) ** 5
    alpha = rough ** 2
    cos_m = torch.einsum('ijk,ik->ij', h, normal)
    chi = (cos_m > 0).float()
    cos_m_sq = cos_m ** 2
    tan_m_sq = _divide_no_nan(1 - cos_m_sq, cos_m_sq)
    d = _divide_no_nan(alpha ** 2 * chi, np_pi * cos_m_sq ** 2 * (alpha ** 2 + tan_m_sq) ** 2)
    cos_v = (normal * pts2c).sum(-1)
    cos_t = torch.einsum('ijk,ik->ij', h, pts2c)
    chi_g = (_divide_no_nan(cos_t, cos_v[:, None].expand_as(cos_t)) > 0).float()
    cos_v_sq = torch.clamp(cos_v ** 2, 0., 1.)
    tan_v_sq = torch.clamp(_divide_no_nan(1 - cos_v_sq, cos_v_sq), min=0.)
    g = _divide_no_nan(chi_g * 2, 1 + torch.sqrt(1 + alpha ** 2 * tan_v_sq[:, None]))
    l_dot_n = torch.einsum('ijk,ik->ij', pts2l, normal)
    spec = _divide_no_nan(f * g * d, 4 * l_dot_n.abs() * cos_v.abs()[:, None])
    return albedo[:, None, :] / np_pi + spec[:, :, None]


np_pi = 3.141592653589793


def _rot(vector, axis, angle):
    axis = torch.tensor(axis, dtype=vector.dtype).reshape(1, 3)
    c, s = torch.cos(angle)[:, None], torch.sin(angle)[:, None]
    return vector * c + axis * (vector @ axis.T) * (1 - c) + torch.cross(axis.expand_as(vector), vector, dim=1) * s


def _dir2rusink(a, b):
    a, b = _safe_l2n(a, 1), _safe_l2n(b, 1)
    h = _safe_l2n((a + b) / 2, 1)
    theta_h = torch.acos(torch.clamp(h[:, 2], -1., 1.))
    phi_h = torch.atan2(h[:, 1], h[:, 0])
    diff = _rot(_rot(b, (0., 0., 1.), -phi_h), (0., 1., 0.), -theta_h)
    theta_d = torch.acos(torch.clamp(diff[:, 2], -1., 1.))
    phi_d = torch.remainder(torch.atan2(diff[:, 1], diff[:, 0]), np_pi)
    return torch.stack((phi_d, theta_h, theta_d), -1)


def _learned_brdf(pts2l, pts2c, normal, albedo, z, brdf_net, brdf_scale, mlp_chunk):
    nrm = _safe_l2n(normal, 1)
    up = torch.tensor((0., 0., 1.)) + 1e-6
    t = _safe_l2n(torch.cross(nrm, up.expand_as(nrm), dim=1), 1)
    b = _safe_l2n(torch.cross(nrm, t, dim=1), 1)
    rot = torch.stack((t, b, nrm), 1)
    vdir = torch.einsum('jkl,jl->jk', rot, pts2c)
    ldir = torch.einsum('jkl,jnl->jnk', rot, pts2l)
    n, nl = ldir.shape[:2]
    ldir_flat = ldir.reshape(-1, 3)
    vdir_flat = vdir[:, None, :].expand(n, nl, 3).reshape(-1, 3)
    rusink = _dir2rusink(ldir_flat, vdir_flat)
    z_flat = z[:, None, :].expand(n, nl, z.shape[1]).reshape(-1, z.shape[1])
    front = ldir_flat[:, 2] > 0
    rz = torch.cat((rusink[front], z_flat[front]), 1)

    def chunk_func(rusink_z):
        x = torch.cat((rusink_z[:, 3:], embed(rusink_z[:, :3], 2)), 1)
        h = mlp(x, brdf_net['brdf_mlp'], ['relu'] * 4, skip_at=[2])
        return mlp(h, brdf_net['brdf_out'], ['softplus'])

    spec = torch.zeros(n * nl, 1)
    spec[front] = _chunk_apply(chunk_func, rz, mlp_chunk)
    return albedo[:, None, :] / np_pi + spec.reshape(n, nl, 1).expand(n, nl, 3) * brdf_scale, float(front.float().mean())


def _linear2srgb(x):
    x = torch.clamp(x, 0., 1.)
    return torch.where(x <= 0.0031308, x * 12.92, 1.055 * torch.pow(x, 1 / 2.4) - 0.055)


def nerfactor_render(batch, net, lxyz, lareas, lights, variant='microfacet', brdf_net=None, f0=0.04, brdf_scale=1.,
                     albedo_slope=0.77, albedo_bias=0.03, to_srgb=True, mlp_chunk=65536):
    """Model.call(mode='test', relight_probes=True) without jitter: batch = (rayo, alpha, xyz) torch fp32 tensors,
    net = {'normal_mlp': [(W, b)], 'normal_out': ..., 'lvis_*', 'albedo_*', 'brdf_z_*'}, lights = [P, L, 3] (the
    trained light first, then the probes).  Returns {'rgb' [N, P, 3], 'normal', 'lvis', 'albedo', 'brdf'} with
    zeros on the alpha = 0 rays (tf.scatter_nd)."""
    rayo, alpha, xyz = batch
    mask = alpha[:, 0] > 0
    rayo_m, xyz_m = rayo[mask], xyz[mask]
    surf2l = _safe_l2n(lxyz.reshape(1, -1, 3) - xyz_m[:, None, :], 2)
    surf2c = _safe_l2n(rayo_m - xyz_m, 1)
    n, nl = surf2l.shape[:2]
    normal = _chunk_apply(lambda p: _mlp128(embed(p, 10), net, 'normal', None), xyz_m, mlp_chunk) + 1e-6
    normal = _safe_l2n(normal, 1)
    rows = torch.cat((xyz_m[:, None, :].expand(n, nl, 3).reshape(-1, 3), surf2l.reshape(-1, 3)), 1)
    lvis = _chunk_apply(lambda r: _mlp128(torch.cat((embed(r[:, :3], 10), embed(r[:, 3:], 4)), -1), net, 'lvis',
                                          'sigmoid'), rows, mlp_chunk).reshape(n, nl)
    albedo = albedo_slope * _chunk_apply(lambda p: _mlp128(embed(p, 10), net, 'albedo', 'sigmoid'), xyz_m,
                                         mlp_chunk) + albedo_bias
    z = _chunk_apply(lambda p: _mlp128(embed(p, 10), net, 'brdf_z', 'sigmoid' if variant == 'microfacet' else None),
                     xyz_m, mlp_chunk)
    front_frac = None
    if variant == 'microfacet':
        brdf = _microfacet(surf2l, surf2c, normal, albedo, z, f0)
    else:
        brdf, front_frac = _learned_brdf(surf2l, surf2c, normal, albedo, z, brdf_net, brdf_scale, mlp_chunk)
    cos = torch.einsum('ijk,ik->ij', surf2l, normal)
    lv = (cos > 0).float() * lvis
    areas = lareas.reshape(1, -1, 1)
    rgbs = []
    for light in lights:
        contrib = brdf * (lv[:, :, None] * light.reshape(1, -1, 3)) * cos[:, :, None] * areas
        rgb = torch.clamp(contrib.sum(1), 0., 1.)
        rgbs.append(_linear2srgb(rgb) if to_srgb else rgb)

    def full(v):
        out = torch.zeros((alpha.shape[0],) + tuple(v.shape[1:]))
        out[mask] = v
        return out

    return {'rgb': full(torch.stack(rgbs, 1)), 'normal': full(normal), 'lvis': full(lvis), 'albedo': full(albedo),
            'brdf': full(z), 'front_lit_frac': front_frac}
