"""NumPy restatement of the NeRFactor surface-shading stage (TEST INFRASTRUCTURE, see
oracle/__init__.py; pinned to the reference's own model code run on a NumPy TF shim,
tests/golden/make_reference_golden.py; the TF kernels themselves stay unpinned).

Follows, op for op:
  brdf/renderer.py:184-219 + xiuminglib/geometry/sph.py:185-190   gen_light_xyz
  nerfactor/models/shape.py:128-144, 196-237                      _calc_ldir/_calc_vdir, _pred_normal_at,
                                                                  _pred_lvis_at
  nerfactor/models/nerfactor.py:181-461                           call, _render, _pred_albedo_at,
                                                                  _pred_brdf_at, _eval_brdf_at
  nerfactor/models/nerfactor_microfacet.py:108-124                sigmoid roughness, Microfacet hook
  brdf/microfacet/microfacet.py:30-111                            Microfacet
  nerfactor/util/geom.py:119-192                                  gen_world2local, dir2rusink
  nerfactor/util/img.py:76-95, 140-163                            alpha_blend, linear2srgb
  nerfactor/models/brdf.py:57-66                                  BRDF MLP shape
  losses: nerfactor.py:463-541, shape.py:239-277, losses.py:32-46
dtype-generic like oracle/nerf_ref.py.
"""
import numpy as np

from .nerf_ref import embed, glorot_uniform, l2_normalize, mlp


# ----------------------------------------------------------------------------- lights
def gen_light_xyz(envmap_h, envmap_w, envmap_radius=1e2):
    """brdf/renderer.py:184-219: lat-long light positions (float64) and solid angles."""
    lat_step = np.pi / (envmap_h + 2)
    lng_step = 2 * np.pi / (envmap_w + 2)
    lats = np.linspace(np.pi / 2 - lat_step, -np.pi / 2 + lat_step, envmap_h)
    lngs = np.linspace(np.pi - lng_step, -np.pi + lng_step, envmap_w)
    lngs, lats = np.meshgrid(lngs, lats)
    r = envmap_radius
    xyz = np.stack((r * np.cos(lats) * np.cos(lngs), r * np.cos(lats) * np.sin(lngs),
                    r * np.sin(lats)), -1)  # sph.py:185-190
    sin_colat = np.sin(np.pi / 2 - lats)
    areas = 4 * np.pi * sin_colat / np.sum(sin_colat)
    return xyz, areas


def one_hot_light(h, w, i, j, inten, ambient, dtype=np.float32):
    """novel_olat entries (nerfactor.py:79-83; util/tensor.py:57-64)."""
    env = np.full((h, w, 3), ambient, dtype)
    env[i, j, :] += np.dtype(dtype).type(inten)
    return env


# ------------------------------------------------------------------------------- nets
def init_mlp128(rng, in_dims, out_dims, width=128, depth=4, skip_at=2, dtype=np.float32):
    """mlp.Network([128]*4, relu, skip_at=[2]) + out layer (shape.py:79-94)."""
    layers, fan_in = [], in_dims
    for i in range(depth):
        layers.append((glorot_uniform(rng, fan_in, width, dtype), np.zeros(width, dtype)))
        fan_in = width + in_dims if i == skip_at else width
    # (a skip behind the LAST body layer: the head reads concat(y, x), mlp.py:47-48 — Keras infers its input width)
    out = [(glorot_uniform(rng, fan_in, out_dims, dtype), np.zeros(out_dims, dtype))]
    return layers, out


def mlp128(x, layers, out, out_act, quant=None, skip_at=2):
    h = mlp(x, layers, ['relu'] * len(layers), skip_at=[skip_at], quant=quant)
    return mlp(h, out, [out_act], quant=quant)


def calc_ldir(pts, lxyz, eps=1e-6):
    """shape.py:128-135: normalize(lxyz[l] - pts[n]) -> [N, L, 3]."""
    return l2_normalize(lxyz.reshape(1, -1, 3).astype(pts.dtype) - pts[:, None, :], 2, eps)


def calc_vdir(cam_loc, pts, eps=1e-6):
    """shape.py:137-144."""
    return l2_normalize(cam_loc - pts, 1, eps)


def pred_normal_at(pts, net, xyz_scale=1., eps=1e-6, quant=None, n_freqs_xyz=10, skip_at=2):
    """shape.py:196-211 (raw, un-normalised, +1e-6).  (n_freqs_xyz / skip_at: config/shape.ini values by default.)"""
    pe = embed((pts.dtype.type(xyz_scale) * pts), n_freqs_xyz)
    return mlp128(pe, net['normal_mlp'], net['normal_out'], None, quant, skip_at) + pts.dtype.type(eps)


def pred_lvis_at(pts, surf2l, net, xyz_scale=1., quant=None, n_freqs_xyz=10, n_freqs_ldir=4, skip_at=2):
    """shape.py:213-237: sigmoid MLP on concat(posenc10(pts), posenc4(ldir)) per (point, light)."""
    n, nl = surf2l.shape[:2]
    surf = np.broadcast_to((pts.dtype.type(xyz_scale) * pts)[:, None, :], (n, nl, 3)).reshape(-1, 3)
    x = np.concatenate((embed(surf, n_freqs_xyz), embed(surf2l.reshape(-1, 3), n_freqs_ldir)), -1)
    return mlp128(x, net['lvis_mlp'], net['lvis_out'], 'sigmoid', quant, skip_at).reshape(n, nl)


def pred_albedo_at(pts, net, xyz_scale=1., slope=0.77, bias=0.03, quant=None):
    """nerfactor.py:377-396."""
    pe = embed((pts.dtype.type(xyz_scale) * pts), 10)
    a = mlp128(pe, net['albedo_mlp'], net['albedo_out'], 'sigmoid', quant)
    return pts.dtype.type(slope) * a + pts.dtype.type(bias)


def pred_brdf_at(pts, net, xyz_scale=1., out_act=None, quant=None):
    """nerfactor.py:398-411 (linear z) / nerfactor_microfacet.py:108-114 (sigmoid roughness)."""
    pe = embed((pts.dtype.type(xyz_scale) * pts), 10)
    return mlp128(pe, net['brdf_z_mlp'], net['brdf_z_out'], out_act, quant)


# --------------------------------------------------------------------------- geometry
def divide_no_nan(a, b):
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(b == 0, np.zeros_like(a * b), a / np.where(b == 0, np.ones_like(b), b))


def gen_world2local(normal, eps=1e-6):
    """util/geom.py:119-149: rows = (tangent, binormal, normal)."""
    dt = normal.dtype.type
    normal = l2_normalize(normal, 1, eps)
    z = np.broadcast_to(np.array((0, 0, 1), normal.dtype) + dt(eps), normal.shape)
    t = l2_normalize(np.cross(normal, z), 1, eps)
    b = l2_normalize(np.cross(normal, t), 1, eps)
    return np.stack((t, b, normal), 1)


def _rot_vec(vector, axis, angle):
    """util/geom.py:168-180 (Rodrigues, fixed axis)."""
    axis = np.asarray(axis, vector.dtype).reshape(1, 3)
    c, s = np.cos(angle)[:, None], np.sin(angle)[:, None]
    return vector * c + axis * (vector @ axis.T) * (1 - c) + \
        np.cross(np.broadcast_to(axis, vector.shape), vector) * s


def dir2rusink(a, b, eps=1e-6):
    """util/geom.py:152-192: (phi_d, theta_h, theta_d)."""
    a = l2_normalize(a, 1, eps)
    b = l2_normalize(b, 1, eps)
    h = l2_normalize((a + b) / 2, 1, eps)
    theta_h = np.arccos(np.clip(h[:, 2], -1., 1.))
    phi_h = np.arctan2(h[:, 1], h[:, 0])
    diff = _rot_vec(_rot_vec(b, (0, 0, 1), -phi_h), (0, 1, 0), -theta_h)
    theta_d = np.arccos(np.clip(diff[:, 2], -1., 1.))
    phi_d = np.mod(np.arctan2(diff[:, 1], diff[:, 0]), a.dtype.type(np.pi))
    return np.stack((phi_d, theta_h, theta_d), -1).astype(a.dtype)


# ------------------------------------------------------------------------------ BRDFs
def microfacet(pts2l, pts2c, normal, albedo, rough, f0=0.04, lambert_only=False):
    """brdf/microfacet/microfacet.py:30-111.  pts2l [N,L,3], pts2c/normal/albedo [N,3], rough [N,1]."""
    dt = pts2c.dtype.type
    pts2l = l2_normalize(pts2l, 2, 1e-6)
    pts2c = l2_normalize(pts2c, 1, 1e-6)
    normal = l2_normalize(normal, 1, 1e-6)
    h = l2_normalize(pts2l + pts2c[:, None, :], 2, 1e-6)
    # Fresnel (Schlick), :106-111
    f = dt(f0) + dt(1 - f0) * (dt(1) - np.einsum('ijk,ijk->ij', pts2l, h)) ** 5
    alpha = rough ** 2
    # D (GGX), :92-104
    cos_m = np.einsum('ijk,ik->ij', h, normal)
    chi = np.where(cos_m > 0, dt(1), dt(0))
    cos_m_sq = cos_m ** 2
    tan_m_sq = divide_no_nan(1 - cos_m_sq, cos_m_sq)
    d = divide_no_nan(alpha ** 2 * chi, dt(np.pi) * cos_m_sq ** 2 * (alpha ** 2 + tan_m_sq) ** 2)
    # G (uses the view direction only), :74-90
    cos_v = np.einsum('ij,ij->i', normal, pts2c)
    cos_t = np.einsum('ijk,ik->ij', h, pts2c)
    chi_g = np.where(divide_no_nan(cos_t, np.broadcast_to(cos_v[:, None], cos_t.shape)) > 0, dt(1), dt(0))
    cos_v_sq = np.clip(cos_v ** 2, 0., 1.)
    tan_v_sq = np.clip(divide_no_nan(1 - cos_v_sq, cos_v_sq), 0., np.inf)
    g = divide_no_nan(chi_g * 2, 1 + np.sqrt(1 + alpha ** 2 * tan_v_sq[:, None]))
    l_dot_n = np.einsum('ijk,ik->ij', pts2l, normal)
    denom = 4 * np.abs(l_dot_n) * np.abs(cos_v)[:, None]
    spec = divide_no_nan(f * g * d, denom)
    brdf = np.broadcast_to((albedo / dt(np.pi))[:, None, :], spec.shape + (3,))
    if not lambert_only:
        brdf = brdf + spec[:, :, None]
    return brdf.astype(pts2c.dtype)


def init_brdf_mlp(rng, z_dim=3, n_freqs=2, dtype=np.float32):
    """models/brdf.py:57-66 at config/brdf.ini: input z_dim + (3+6*2)."""
    layers, out = init_mlp128(rng, z_dim + 3 + 6 * n_freqs, 1, dtype=dtype)
    return {'brdf_mlp': layers, 'brdf_out': out}


def learned_spec(pts2l, pts2c, normal, z, brdf_net, n_freqs_rusink=2, quant=None):
    """The achromatic specular term of Model._eval_brdf_at (nerfactor.py:413-458): 0 for
    back-lit directions (local l.z <= 0), softplus(MLP([z, posenc(rusink)])) otherwise."""
    n, nl = pts2l.shape[:2]
    rot = gen_world2local(normal)
    vdir = np.einsum('jkl,jl->jk', rot, pts2c)
    ldir = np.einsum('jkl,jnl->jnk', rot, pts2l)
    ldir_flat = ldir.reshape(-1, 3)
    vdir_flat = np.broadcast_to(vdir[:, None, :], ldir.shape).reshape(-1, 3)
    rusink = dir2rusink(ldir_flat, vdir_flat)
    z_flat = np.broadcast_to(z[:, None, :], (n, nl, z.shape[1])).reshape(-1, z.shape[1])
    front = ldir_flat[:, 2] > 0
    x = np.concatenate((z_flat, embed(rusink, n_freqs_rusink)), 1)
    spec = np.zeros(n * nl, pts2l.dtype)
    if np.any(front):
        y = mlp(x[front], brdf_net['brdf_mlp'], ['relu'] * 4, skip_at=[2], quant=quant)
        y = mlp(y, brdf_net['brdf_out'], ['softplus'], quant=quant)
        spec[front] = y[:, 0]
    return spec.reshape(n, nl)


def learned_brdf(pts2l, pts2c, normal, albedo, z, brdf_net, brdf_scale=1., quant=None):
    """nerfactor.py:459-461."""
    spec = learned_spec(pts2l, pts2c, normal, z, brdf_net, quant=quant)
    dt = albedo.dtype.type
    return albedo[:, None, :] / dt(np.pi) + spec[:, :, None] * dt(brdf_scale)


# ---------------------------------------------------------------------------- rendering
def linear2srgb(x):
    """util/img.py:140-163."""
    x = np.clip(x, 0, 1)
    return np.where(x <= 0.0031308, x * 12.92, 1.055 * np.power(x, 1 / 2.4) - 0.055).astype(x.dtype)


def integrate(brdf, lvis, surf2l, normal, light, areas, to_srgb=True):
    """Model._render.integrate (nerfactor.py:325-342) for one light [h,w,3] / [L,3]."""
    cos = np.einsum('ijk,ik->ij', surf2l, normal)
    lv = (cos > 0).astype(brdf.dtype) * lvis
    lf = light.reshape(-1, 3).astype(brdf.dtype)
    contrib = brdf * (lv[:, :, None] * lf[None]) * cos[:, :, None] * \
        areas.reshape(1, -1, 1).astype(brdf.dtype)
    rgb = np.clip(np.sum(contrib, 1), 0., 1.)
    return linear2srgb(rgb) if to_srgb else rgb


# --------------------------------------------------------------------------- full model
def init_nerfactor_net(rng, z_dim, dtype=np.float32):
    net = {}
    net['normal_mlp'], net['normal_out'] = init_mlp128(rng, 63, 3, dtype=dtype)
    net['lvis_mlp'], net['lvis_out'] = init_mlp128(rng, 90, 1, dtype=dtype)
    net['albedo_mlp'], net['albedo_out'] = init_mlp128(rng, 63, 3, dtype=dtype)
    net['brdf_z_mlp'], net['brdf_z_out'] = init_mlp128(rng, 63, z_dim, dtype=dtype)
    return net


def nerfactor_call(batch, net, lxyz, lareas, light, variant='microfacet', brdf_net=None, f0=0.04,
                   brdf_scale=1., albedo_slope=0.77, albedo_bias=0.03, xyz_scale=1.,
                   to_srgb=True, probes=None, olat=None, xyz_noise=None, normalize_z=False,
                   quant=None, shape_mode='finetune', albedo_scales=None, albedo_override=None,
                   brdf_z_override=None):
    """Model.call (nerfactor.py:181-313), shape_mode in (scratch, finetune, frozen).
    batch = (rayo, rgb, alpha, xyz, normal, lvis) flattened [N,...].
    probes: list of [h,w,3] lights -> rgb_probes [N,P,3]; olat: (inten, ambient) -> rgb_olat [N,L,3].
    xyz_noise: the jitter tensor for the masked points (tf.random.normal stand-in) or None.
    shape_mode 'nerf' (:204-207, :214-216): normals / visibility are the NeRF-derived buffers of the batch.
    albedo_scales [3], albedo_override [3] or [N,3], brdf_z_override [z_dim]: the editing hooks of :236-261."""
    rayo, rgb, alpha, xyz, normal, lvis = batch
    dt = xyz.dtype.type
    mask = alpha[:, 0] > 0
    rayo_m, xyz_m = rayo[mask], xyz[mask]
    surf2l = calc_ldir(xyz_m, lxyz)
    surf2c = calc_vdir(rayo_m, xyz_m)
    out_act = 'sigmoid' if variant == 'microfacet' else None

    def heads(p):
        nrm = l2_normalize(pred_normal_at(p, net, xyz_scale, quant=quant), 1, 1e-6)
        lv = pred_lvis_at(p, surf2l, net, xyz_scale, quant=quant)
        alb = pred_albedo_at(p, net, xyz_scale, albedo_slope, albedo_bias, quant=quant)
        zz = pred_brdf_at(p, net, xyz_scale, out_act, quant=quant)
        if normalize_z:
            zz = l2_normalize(zz, 1, 1e-6)
        return nrm, lv, alb, zz

    normal_pred, lvis_pred, albedo, brdf_prop = heads(xyz_m)
    jit = heads(xyz_m + xyz_noise) if xyz_noise is not None else (None,) * 4
    if shape_mode == 'nerf':
        normal_pred = l2_normalize(normal[mask], 1, 1e-6)
        lvis_pred = np.clip(lvis[mask], dt(1e-8), dt(1.))
        jit = (None, None) + tuple(jit[2:])
    if albedo_scales is not None:
        albedo = np.asarray(albedo_scales, xyz.dtype).reshape(1, 3) * albedo
    if albedo_override is not None:
        ao = np.asarray(albedo_override, xyz.dtype)
        albedo = np.tile(ao[None, :], (albedo.shape[0], 1)) if ao.ndim == 1 else ao[mask]
    if brdf_z_override is not None:
        brdf_prop = np.tile(np.asarray(brdf_z_override, xyz.dtype).reshape(1, -1), (brdf_prop.shape[0], 1))
    if variant == 'microfacet':
        brdf = microfacet(surf2l, surf2c, normal_pred, albedo, brdf_prop, f0=f0)
    else:
        brdf = learned_brdf(surf2l, surf2c, normal_pred, albedo, brdf_prop, brdf_net, brdf_scale,
                            quant=quant)
    rgb_pred = integrate(brdf, lvis_pred, surf2l, normal_pred, light, lareas, to_srgb)
    n, nl = alpha.shape[0], lvis_pred.shape[1]

    def scatter(v, shape):
        if v is None:
            return None
        full = np.zeros((n,) + shape, v.dtype)
        full[mask] = v
        return full

    pred = {'rgb': scatter(rgb_pred, (3,)), 'normal': scatter(normal_pred, (3,)),
            'lvis': scatter(lvis_pred, (nl,)), 'albedo': scatter(albedo, (3,)),
            'brdf': scatter(brdf_prop, (brdf_prop.shape[1],))}
    if probes is not None:
        rp = np.stack([integrate(brdf, lvis_pred, surf2l, normal_pred, p, lareas, to_srgb)
                       for p in probes], 1)
        pred['rgb_probes'] = scatter(rp, rp.shape[1:])
    if olat is not None:
        inten, ambient = olat
        h, w = lareas.shape
        ro = np.stack([integrate(brdf, lvis_pred, surf2l, normal_pred,
                                 one_hot_light(h, w, i, j, inten, ambient, xyz.dtype), lareas, to_srgb)
                       for i in range(h) for j in range(w)], 1)
        pred['rgb_olat'] = scatter(ro, ro.shape[1:])
    gt = {'rgb': scatter(rgb[mask], (3,)), 'normal': scatter(normal[mask], (3,)),
          'lvis': scatter(lvis[mask], (nl,)), 'alpha': alpha}
    loss_kwargs = {'normal_jitter': scatter(jit[0], (3,)), 'lvis_jitter': scatter(jit[1], (nl,)),
                   'albedo_jitter': scatter(jit[2], (3,)),
                   'brdf_prop_jitter': scatter(jit[3], (brdf_prop.shape[1],))}
    aux = {'brdf': brdf, 'surf2l': surf2l, 'surf2c': surf2c, 'mask': mask}
    return pred, gt, loss_kwargs, aux


def nerfactor_loss(pred, gt, loss_kwargs, light, mode='train', white_bg=True, shape_trainable=True,
                   normal_loss_weight=0.1, lvis_loss_weight=0.1, normal_smooth_weight=0.05,
                   lvis_smooth_weight=0.05, albedo_smooth_weight=0.05, brdf_smooth_weight=0.01,
                   light_tv_weight=5e-6, light_achro_weight=0., smooth_use_l1=True):
    """Model.compute_loss (nerfactor.py:463-541), per-ray (keep_batch) + the scalar light prior."""
    alpha = gt['alpha']
    dt = alpha.dtype.type
    bgv = dt(1.) if white_bg else dt(0.)

    def blend(x):
        return x * alpha + bgv * (dt(1.) - alpha)

    def mse(a, b):
        return np.mean((a - b) ** 2, -1)

    def smooth(a, b):
        return np.mean(np.abs(a - b), -1) if smooth_use_l1 else mse(a, b)

    rgb_pred, rgb_gt = blend(pred['rgb']), blend(gt['rgb'])
    normal_pred, normal_gt = blend(pred['normal']), blend(gt['normal'])
    lvis_pred, lvis_gt = blend(pred['lvis']), blend(gt['lvis'])
    loss = mse(rgb_gt, rgb_pred)
    if mode == 'vali':
        return loss
    if shape_trainable:
        loss = loss + dt(normal_loss_weight) * mse(normal_gt, normal_pred)
        loss = loss + dt(lvis_loss_weight) * mse(lvis_gt, lvis_pred)
        if loss_kwargs.get('normal_jitter') is not None:
            loss = loss + dt(normal_smooth_weight) * smooth(normal_pred, loss_kwargs['normal_jitter'])
        if loss_kwargs.get('lvis_jitter') is not None:
            loss = loss + dt(lvis_smooth_weight) * smooth(lvis_pred, loss_kwargs['lvis_jitter'])
    if loss_kwargs.get('albedo_jitter') is not None:
        loss = loss + dt(albedo_smooth_weight) * smooth(pred['albedo'], loss_kwargs['albedo_jitter'])
    if loss_kwargs.get('brdf_prop_jitter') is not None:
        loss = loss + dt(brdf_smooth_weight) * smooth(pred['brdf'], loss_kwargs['brdf_prop_jitter'])
    if mode == 'train':
        if light_tv_weight > 0:
            dx = light - np.roll(light, 1, 1)
            dy = light - np.roll(light, 1, 0)
            loss = loss + dt(light_tv_weight) * np.sum(dx ** 2 + dy ** 2)
        if light_achro_weight > 0:
            dc = light - np.roll(light, 1, 2)
            loss = loss + dt(light_achro_weight) * np.sum(dc ** 2)
    return loss


# ------------------------------------------------------------------- shape pre-training model
def shape_call(xyz, net, lxyz, xyz_scale=1., xyz_noise=None, quant=None):
    """models/shape.py:146-182 (Model.call): unit normals and light visibility at the surface points, plus the
    jittered second evaluation when xyz_noise (the tf.random.normal stand-in) is given."""
    surf2l = calc_ldir(xyz, lxyz)

    def heads(p):
        return (l2_normalize(pred_normal_at(p, net, xyz_scale, quant=quant), 1, 1e-6),
                pred_lvis_at(p, surf2l, net, xyz_scale, quant=quant))

    normal, lvis = heads(xyz)
    jit = heads(xyz + xyz_noise) if xyz_noise is not None else (None, None)
    return {'normal': normal, 'lvis': lvis}, {'normal_jitter': jit[0], 'lvis_jitter': jit[1]}


def shape_loss(pred, gt, loss_kwargs, white_bg=True, normal_loss_weight=1., lvis_loss_weight=1.,
               normal_smooth_weight=0., lvis_smooth_weight=0., smooth_use_l1=True):
    """models/shape.py:239-281 (Model.compute_loss): per-point loss on alpha-composited predictions."""
    alpha = gt['alpha']
    dt = alpha.dtype.type
    bgv = dt(1.) if white_bg else dt(0.)

    def blend(x):
        return x * alpha + bgv * (dt(1.) - alpha)

    def mse(a, b):
        return np.mean((a - b) ** 2, -1)

    def smooth(a, b):
        return np.mean(np.abs(a - b), -1) if smooth_use_l1 else mse(a, b)

    normal_pred, lvis_pred = blend(pred['normal']), blend(pred['lvis'])
    loss = dt(normal_loss_weight) * mse(blend(gt['normal']), normal_pred)
    loss = loss + dt(lvis_loss_weight) * mse(blend(gt['lvis']), lvis_pred)
    if loss_kwargs.get('normal_jitter') is not None:
        loss = loss + dt(normal_smooth_weight) * smooth(normal_pred, loss_kwargs['normal_jitter'])
    if loss_kwargs.get('lvis_jitter') is not None:
        loss = loss + dt(lvis_smooth_weight) * smooth(lvis_pred, loss_kwargs['lvis_jitter'])
    return loss


# ------------------------------------------------------------------------- BRDF prior model
def brdf_prior_eval(z, rusink, brdf_net, n_freqs=2, quant=None):
    """models/brdf.py:101-123 (Model._eval_brdf_at): softplus MLP on [z, posenc(rusink)] and on the reciprocal
    configuration (phi_d + pi).  z [N, z_dim], rusink [N, 3] -> (brdf, brdf_reci), both [N, 1]."""
    def run(r):
        x = np.concatenate((z, embed(r, n_freqs)), 1)
        y = mlp(x, brdf_net['brdf_mlp'], ['relu'] * 4, skip_at=[2], quant=quant)
        return mlp(y, brdf_net['brdf_out'], ['softplus'], quant=quant)
    reci = np.concatenate((rusink[:, :1] + rusink.dtype.type(np.pi), rusink[:, 1:]), 1)
    return run(rusink), run(reci)


def latent_interp(z_table, w1, i1, w2, i2):
    """networks/layers.py:57-68 (LatentCode.interp), un-normalised codes: linear interpolation -> [1, z_dim]."""
    dt = z_table.dtype.type
    return dt(w1) * z_table[i1:i1 + 1] + dt(w2) * z_table[i2:i2 + 1]


def brdf_prior_loss(refl, brdf, brdf_reci, transform='log'):
    """models/brdf.py:125-141 (Model.compute_loss) with loss = l2 (losses.py:32-46, keep_batch=False): the scalar
    mean squared error of f(prediction) against f(measurement), summed over the two reciprocal halves."""
    f = {'log': np.log, 'none': lambda x: x, 'divide': lambda x: x / (x + 1)}[transform.lower()]
    return np.mean((f(refl) - f(brdf)) ** 2) + np.mean((f(refl) - f(brdf_reci)) ** 2)
