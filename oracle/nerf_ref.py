"""NumPy restatement of the vanilla-NeRF ray-marching stage (TEST INFRASTRUCTURE, see
oracle/__init__.py; pinned to the reference's own model code run on a NumPy TF shim,
tests/golden/make_reference_golden.py; the TF kernels themselves stay unpinned).

Follows, op for op:
  nerfactor/models/nerf.py:120-290   gen_z, gen_z_fine, _render_rays, accumulate_sigma,
                                     _accumulate, _eval_nerf_at
  nerfactor/util/math.py:63-94       safe_l2_normalize, safe_cumprod, inv_transform_sample
  nerfactor/networks/embedder.py:23-47, mlp.py:24-50, seq.py:33-38
  nerfactor/util/img.py:76-95        alpha_blend
  nerfactor/datasets/nerf.py:172-193 _gen_rays (benchmark-input generator)

All functions are dtype-generic: pass float32 arrays to mirror TF's compute type, float64
for the anchor.  ``quant`` (optional) rounds every matmul operand the way the bf16 MFMA
path does (weights and layer inputs to bf16, fp32 accumulate) so kernel logic can be
checked tightly, independent of quantisation noise.
"""
import numpy as np


# ----------------------------------------------------------------------------- helpers
def bf16_round(x):
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (what v_cvt_pk_bf16_f32 does)."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    u = (u + (np.uint32(0x7FFF) + ((u >> np.uint32(16)) & np.uint32(1)))) & np.uint32(0xFFFF0000)
    return u.view(np.float32).reshape(x.shape)


def bf16_pair_round(x):
    """The operand the fp32-class kernels see (mlp_x3.hpp: hi = bf16(x), lo = bf16(x - hi)): hi + lo, 16 significant
    bits.  With it as `quant`, an oracle MLP reproduces those kernels up to the dropped lo x lo products (2^-18
    relative) and fp32 summation order."""
    x = np.ascontiguousarray(x, dtype=np.float32)
    hi = bf16_round(x)
    return hi + bf16_round(x - hi)


def l2_normalize(x, axis, eps):
    """tf.linalg.l2_normalize: x * rsqrt(max(sum(x^2), eps)) (nerf.py:157 eps=1e-12;
    util/math.py:63-64 eps=1e-6)."""
    sq = np.sum(x * x, axis=axis, keepdims=True)
    return x * (1. / np.sqrt(np.maximum(sq, x.dtype.type(eps))))


def glorot_uniform(rng, fan_in, fan_out, dtype=np.float32):
    """Keras Dense default kernel init (mlp.py:35): U(-l, l), l = sqrt(6/(in+out))."""
    lim = np.sqrt(6. / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(dtype)


# ------------------------------------------------------------------- networks/embedder.py
def embed(x, n_freqs):
    """Embedder.__call__ (embedder.py:23-47) with incl_input, log_sampling and
    periodic_func=[sin, cos]: concat [x, sin(f0 x), cos(f0 x), sin(f1 x), cos(f1 x), ...],
    f_k = 2^k (2.**linspace(0, L-1, L)).  out dims 3 + 6L."""
    outs = [x]
    for k in range(n_freqs):
        f = x.dtype.type(2. ** k)
        outs.append(np.sin(x * f))
        outs.append(np.cos(x * f))
    return np.concatenate(outs, -1)


# ------------------------------------------------------------------------ networks/mlp.py
def _act(y, name):
    if name is None:
        return y
    if name == 'relu':
        return np.maximum(y, 0, out=y)  # y is always a fresh matmul result here
    if name == 'sigmoid':
        return 1. / (1. + np.exp(-y))
    if name == 'softplus':
        return np.logaddexp(y, 0).astype(y.dtype)
    raise ValueError(name)


def dense(x, kernel, bias, act=None, quant=None):
    """keras Dense: act(x @ kernel[in,out] + bias)."""
    if quant is not None:
        y = quant(x) @ quant(kernel) + bias
    else:
        y = x @ kernel + bias
    return _act(y.astype(x.dtype), act)


def mlp(x, layers, acts, skip_at=None, quant=None):
    """mlp.Network.__call__ (mlp.py:39-50): after layer i in skip_at the ORIGINAL input is
    concatenated behind the activation, (y, x) order, and feeds layer i+1."""
    x_ = x
    y = x
    for i, (kernel, bias) in enumerate(layers):
        y = dense(x_, kernel, bias, acts[i], quant)
        if skip_at is not None and i in skip_at:
            y = np.concatenate((y, x), -1)
        x_ = y
    return y


# ---------------------------------------------------------------------------- NeRF nets
NERF_WIDTH, NERF_DEPTH = 256, 8


def init_nerf_net(rng, n_freqs_xyz=10, n_freqs_view=4, width=NERF_WIDTH, depth=NERF_DEPTH,
                  sigma_bias=0., sigma_gain=1., dtype=np.float32, use_views=True):
    """Random weights with the shapes of Model._init_net (nerf.py:53-71): glorot-uniform
    kernels, zero biases (Keras defaults).  sigma_bias / sigma_gain make the "opaque"
    variant of SURVEY.md §8d (non-trivial opacity exercises compositing + resampling)."""
    dx, dv = 3 + 6 * n_freqs_xyz, 3 + 6 * n_freqs_view
    skip = depth // 2
    enc, fan_in = [], dx
    for i in range(depth):
        enc.append((glorot_uniform(rng, fan_in, width, dtype), np.zeros(width, dtype)))
        fan_in = width + dx if i == skip else width
    d_enc = fan_in      # (enc_depth = 2: the skip sits behind the last layer and the heads read concat(y, embed(x)), mlp.py:47-48)
    if not use_views:   # nerf.py:62-66: one linear head for (rgb, sigma)
        k = glorot_uniform(rng, d_enc, 4, dtype)
        k[:, 3] *= dtype(sigma_gain)
        return {'enc': enc, 'rgbs_out': [(k, np.array([0., 0., 0., sigma_bias], dtype))]}
    net = {
        'enc': enc,
        'sigma_out': [(glorot_uniform(rng, d_enc, 1, dtype) * dtype(sigma_gain),
                       np.full(1, sigma_bias, dtype))],
        'bottleneck': [(glorot_uniform(rng, d_enc, width, dtype), np.zeros(width, dtype))],
        'rgb_out': [(glorot_uniform(rng, width + dv, width // 2, dtype),
                     np.zeros(width // 2, dtype)),
                    (glorot_uniform(rng, width // 2, 3, dtype), np.zeros(3, dtype))],
    }
    return net


def randomize_biases(net, rng, scale=0.1):
    """Non-zero biases so a bias-permutation bug cannot hide behind Keras' zero init."""
    for layers in net.values():
        for i, (k, b) in enumerate(layers):
            layers[i] = (k, rng.uniform(-scale, scale, size=b.shape).astype(b.dtype))
    return net


def eval_nerf_at(pts, views, net, n_freqs_xyz=10, n_freqs_view=4, quant=None,
                 mlp_chunk=65536):
    """Model._eval_nerf_at (nerf.py:256-290); a net with 'rgbs_out' takes the use_views = False branch.
    pts, views [N,S,3] -> rgbs [N,S,4] = concat(raw rgb, raw sigma)."""
    depth = len(net['enc'])
    pts_flat = pts.reshape(-1, 3)
    views_flat = views.reshape(-1, 3)
    chunks = []
    for i in range(0, pts_flat.shape[0], mlp_chunk):
        pe = embed(pts_flat[i:i + mlp_chunk], n_freqs_xyz)
        ve = embed(views_flat[i:i + mlp_chunk], n_freqs_view)
        feat = mlp(pe, net['enc'], ['relu'] * depth, skip_at=[depth // 2], quant=quant)
        if 'rgbs_out' in net:   # use_views = False (nerf.py:283-286)
            chunks.append(mlp(feat, net['rgbs_out'], [None], quant=quant))
            continue
        sigma = mlp(feat, net['sigma_out'], [None], quant=quant)
        feat = mlp(feat, net['bottleneck'], [None], quant=quant)
        rgb = mlp(np.concatenate((feat, ve), -1), net['rgb_out'], ['relu', None],
                  quant=quant)
        chunks.append(np.concatenate([rgb, sigma], -1))
    return np.concatenate(chunks, 0).reshape(pts.shape[:2] + (4,))


# ------------------------------------------------------------------------------ sampling
def linspace01(n, dtype):
    """tf.linspace(0., 1., n) as the TF 2.2 CPU kernel computes it: start + step * i with
    step = (stop - start) / (n - 1) evaluated in the output dtype."""
    dtype = np.dtype(dtype).type
    if n == 1:
        return np.zeros(1, dtype)          # tf.linspace(start, stop, 1) == [start]
    step = dtype(1.) / dtype(n - 1)
    return (np.arange(n).astype(dtype) * step).astype(dtype)


def gen_z(near, far, n_samples, n_rays, lin_in_disp=False, u=None, dtype=np.float32):
    """Model.gen_z (nerf.py:120-136).  u: None (perturb=False) or [n_rays, n_samples]
    uniforms standing in for tf.random.uniform."""
    dt = np.dtype(dtype).type
    t = linspace01(n_samples, dtype)
    near, far = dt(near), dt(far)
    if lin_in_disp:
        z = dt(1.) / (dt(1.) / near * (dt(1.) - t) + dt(1.) / far * t)
    else:
        z = near * (dt(1.) - t) + far * t
    z = np.broadcast_to(z, (n_rays, n_samples)).astype(dtype)
    if u is not None:
        mid = dt(.5) * (z[:, 1:] + z[:, :-1])
        upper = np.concatenate([mid, z[:, -1:]], -1)
        lower = np.concatenate([z[:, :1], mid], -1)
        z = lower + (upper - lower) * u.astype(dtype)
    return z


def seq_cumsum(x):
    """Left-to-right running sum in the array's own dtype (np.cumsum is sequential)."""
    return np.cumsum(x, axis=-1, dtype=x.dtype)


def inv_transform_sample(val, weights, n_samples, u=None, eps=1e-5):
    """util/math.py:71-94.  val [N,B+1] bin positions, weights [N,B].
    u: None => det (linspace), else [N, n_samples]."""
    dt = val.dtype.type
    denom = seq_cumsum(weights)[:, -1:] + dt(eps)
    pdf = weights / denom
    cdf = seq_cumsum(pdf)
    cdf = np.concatenate((np.zeros_like(cdf[:, :1]), cdf), -1)           # [N, B+1]
    if u is None:
        u = np.broadcast_to(linspace01(n_samples, val.dtype), (cdf.shape[0], n_samples))
    u = u.astype(val.dtype)
    # tf.searchsorted(side='right'): number of cdf entries <= u
    ind = np.sum(cdf[:, None, :] <= u[:, :, None], -1).astype(np.int32)
    below = np.maximum(0, ind - 1)
    above = np.minimum(ind, cdf.shape[-1] - 1)
    cdf_b = np.take_along_axis(cdf, below, 1)
    cdf_a = np.take_along_axis(cdf, above, 1)
    val_b = np.take_along_axis(val, below, 1)
    val_a = np.take_along_axis(val, above, 1)
    den = cdf_a - cdf_b
    den = np.where(den < dt(eps), np.ones_like(den), den)
    t = (u - cdf_b) / den
    return val_b + t * (val_a - val_b)


def gen_z_fine(z_coarse, weights, n_samples_fine, u=None):
    """Model.gen_z_fine (nerf.py:138-147)."""
    dt = z_coarse.dtype.type
    mid = dt(.5) * (z_coarse[:, 1:] + z_coarse[:, :-1])
    z_fine = inv_transform_sample(mid, weights[:, 1:-1], n_samples_fine, u=u)
    return np.sort(np.concatenate((z_coarse, z_fine), -1), -1)


# --------------------------------------------------------------------------- compositing
def accumulate_sigma(sigma, z, rayd, noise=None, inf=1e10):
    """Model.accumulate_sigma (nerf.py:184-212) + safe_cumprod (util/math.py:67-68).
    noise: None or the already-scaled N(0,1)*noise_std tensor."""
    dt = z.dtype.type
    dist = z[:, 1:] - z[:, :-1]
    dist = np.concatenate((dist, np.full_like(dist[:, :1], inf)), -1)
    dist = dist * np.sqrt(np.sum(rayd * rayd, -1))[:, None]
    s = sigma if noise is None else sigma + noise
    with np.errstate(over='ignore'):
        density = dt(1.) - np.exp(-np.maximum(s, 0) * dist)
    x = dt(1.) - density + dt(1e-6)
    # tf.math.cumprod(exclusive=True): sequential running product
    excl = np.concatenate((np.ones_like(x[:, :1]),
                           np.cumprod(x[:, :-1], axis=-1, dtype=x.dtype)), -1)
    return density * excl


def accumulate(rgbs, z, rayd, white_bg=True, noise=None, eps=1e-10):
    """Model._accumulate (nerf.py:214-254).  Note rgb is multiplied by occu a second time
    by alpha_blend (util/img.py:95) — reference behaviour, kept."""
    dt = z.dtype.type
    weights = accumulate_sigma(rgbs[:, :, 3], z, rayd, noise=noise)
    rgb = 1. / (1. + np.exp(-rgbs[:, :, :3]))
    occu = np.sum(weights, -1)
    rgb = np.sum(weights[:, :, None] * rgb, -2)
    depth = np.sum(weights * z, -1)
    disp = dt(1.) / np.maximum(depth, dt(eps))
    bg = np.ones_like(rgb) if white_bg else np.zeros_like(rgb)
    rgb = rgb * occu[:, None] + bg * (dt(1.) - occu[:, None])
    return (rgb.astype(z.dtype), occu.astype(z.dtype), depth.astype(z.dtype),
            disp.astype(z.dtype), weights.astype(z.dtype))


# ------------------------------------------------------------------------- full pipeline
def render_rays(rayo, rayd, net_coarse, net_fine, near=2., far=6., n_samples_coarse=64,
                n_samples_fine=128, lin_in_disp=False, white_bg=True, u_coarse=None,
                u_fine=None, quant=None, n_freqs_xyz=10, n_freqs_view=4):
    """Model._render_rays (nerf.py:149-182).  Returns (pred_coarse, pred_fine, aux)."""
    rayd = l2_normalize(rayd, 1, 1e-12)
    n = rayo.shape[0]
    z = gen_z(near, far, n_samples_coarse, n, lin_in_disp, u_coarse, rayo.dtype)
    pts = rayo[:, None, :] + rayd[:, None, :] * z[:, :, None]
    views = np.broadcast_to(rayd[:, None, :], pts.shape)
    rgbs_c = eval_nerf_at(pts, views, net_coarse, n_freqs_xyz, n_freqs_view, quant)
    rgb, occu, depth, disp, weights = accumulate(rgbs_c, z, rayd, white_bg)
    coarse = {'rgb': rgb, 'occu': occu, 'depth': depth, 'disp': disp}
    aux = {'z_coarse': z, 'rgbs_coarse': rgbs_c, 'weights_coarse': weights, 'rayd': rayd}
    if n_samples_fine <= 0:
        return coarse, {}, aux
    z_all = gen_z_fine(z, weights, n_samples_fine, u=u_fine)
    pts = rayo[:, None, :] + rayd[:, None, :] * z_all[:, :, None]
    views = np.broadcast_to(rayd[:, None, :], pts.shape)
    rgbs_f = eval_nerf_at(pts, views, net_fine, n_freqs_xyz, n_freqs_view, quant)
    rgb, occu, depth, disp, w_f = accumulate(rgbs_f, z_all, rayd, white_bg)
    fine = {'rgb': rgb, 'occu': occu, 'depth': depth, 'disp': disp}
    aux.update({'z_all': z_all, 'rgbs_fine': rgbs_f, 'weights_fine': w_f})
    return coarse, fine, aux


def nerf_loss(gt, coarse_rgb, fine_rgb=None):
    """Model.compute_loss (nerf.py:292-300) with loss='l2', keep_batch=True
    (losses.py:32-46): per-ray mean squared error, coarse + fine."""
    loss = np.mean((gt - coarse_rgb) ** 2, -1)
    if fine_rgb is not None:
        loss = loss + np.mean((gt - fine_rgb) ** 2, -1)
    return loss


# ------------------------------------------------------------------- datasets/nerf.py rays
def gen_rays(cam_to_world, angle_x, imh, imw, sps=1):
    """Dataset._gen_rays (datasets/nerf.py:172-193), ndc=False.  float64 in, float64 out
    (the caller casts to float32 like datasets/nerf.py:151)."""
    cam_loc = cam_to_world[:3, 3]
    rayo = np.tile(cam_loc[None, None, :], (imh * sps, imw * sps, 1))
    xs = np.linspace(0, imw, imw * sps, endpoint=False)
    ys = np.linspace(0, imh, imh * sps, endpoint=False)
    xs, ys = np.meshgrid(xs, ys)
    fl = .5 * imw / np.tan(.5 * angle_x)
    rayd = np.stack(((xs - .5 * imw) / fl, -(ys - .5 * imh) / fl, -np.ones_like(xs)), -1)
    rayd = np.sum(rayd[:, :, None, :] * cam_to_world[:3, :3], -1)
    return rayo, rayd


def lookat_cam_to_world(cam_loc, target=(0., 0., 0.), up=(0., 0., 1.)):
    """Blender-convention camera (looks down -z, +y up) on a sphere, for synthetic views
    shaped like the NeRF-synthetic metadata (cam_transform_mat)."""
    cam_loc = np.asarray(cam_loc, np.float64)
    fwd = np.asarray(target, np.float64) - cam_loc
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    upv = np.cross(right, fwd)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, upv, -fwd, cam_loc
    return m


def psnr_uint8_luma(im1, im2):
    """xiuminglib.metric.PSNR('uint8') semantics (metric.py:103-149) on float [0,1] images
    quantised by truncation like xiuminglib/io/img.py:150: luma 0.2126/0.7152/0.0722."""
    def q(x):
        return (np.clip(x, 0, 1) * 255).astype(np.uint8).astype(np.float64)
    w = np.array([0.2126, 0.7152, 0.0722])
    l1, l2 = q(im1) @ w, q(im2) @ w
    mse = np.mean((l1 - l2) ** 2)
    return np.inf if mse == 0 else 10 * np.log10(255. ** 2 / mse)
