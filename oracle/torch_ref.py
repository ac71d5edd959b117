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
    mid = .5 * (z[:, 1:] + z[:, :-1])
    z_fine = inv_transform_sample(mid, w[:, 1:-1], n_fine)
    z_all, _ = torch.sort(torch.cat((z, z_fine), -1), -1)
    pts = rayo[:, None, :] + rayd[:, None, :] * z_all[:, :, None]
    views = rayd[:, None, :].expand(pts.shape)
    rgbs = eval_nerf_at(pts, views, net_fine)
    rgb_f, occu_f, depth_f, _, _ = accumulate(rgbs, z_all, rayd, white_bg)
    return {'rgb': rgb_c, 'occu': occu_c, 'depth': depth_c}, \
           {'rgb': rgb_f, 'occu': occu_f, 'depth': depth_f}, {'z_all': z_all}
