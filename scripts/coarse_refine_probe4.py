"""Which coarse samples must be fp32-class for the bf16 render to hold max-abs 3e-2 on EVERY ray of a fitted NeRF
?  r04 (scripts/fitted_outliers.py): the rays above 3e-2 are a coarse-pass effect — an fp32-class
coarse pass removes them all, at +67 % frame time.  Here: the bf16 coarse pass, then ONLY selected coarse samples re-evaluated
with the fp32-class density kernel before compositing / inverse-CDF sampling.  Selection from the bf16 pass's own outputs:
a sample is re-evaluated when it is VISIBLE (transmittance T_i > t_min) and PARTIALLY transparent (a_lo < alpha_i < a_hi) —
the samples whose density error moves the weights — optionally dilated by its neighbours.  Reports, per criterion: the fraction
of coarse samples selected (cost ~ 3.7 x that fraction of the coarse pass), rays above 3e-2 and max-abs against the
fp32-class render of the same rays (7e-4 from the fp32 CPU oracle), on the fitted weights and on the bench's glorot weights."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import ops, synth  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402

dev = torch.device('cuda:0')
N = int(os.environ.get('PROBE_RAYS', 131072))


def render(o, d, blobs, gblob, prec_c, prec_f, select=None, stats=None):
    z = ops.gen_z(2., 6., 64, o.shape[0], device=dev)
    raw = ops.nerf_mlp_fwd(o, d, z, blobs[prec_c][0], prec_c)
    if prec_c == 'bf16':
        ops.nerf_refine_last_sample(o, d, z, raw, gblob[0])
        if select is not None:
            sig = torch.relu(raw[..., 3])
            dist = torch.cat((z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)), 1) * d.norm(dim=1, keepdim=True)
            alpha = 1 - torch.exp(-sig * dist)
            T = torch.cumprod(torch.cat((torch.ones_like(alpha[:, :1]), 1 - alpha[:, :-1] + 1e-6), 1), 1)
            m = select(alpha, T, raw[..., 3])
            m[:, -1] = False                       # already fp32-class
            ray, smp = torch.nonzero(m, as_tuple=True)
            if stats is not None:
                stats['selected_frac'] = float(m.float().mean())
                stats['rays_touched_frac'] = float(m.any(1).float().mean())
            if ray.numel():
                s32 = ops.nerf_sigma_fwd(o[ray].contiguous(), d[ray].contiguous(), z[ray, smp][:, None].contiguous(), gblob[0], 'fp32')
                raw[ray, smp, 3] = s32[:, 0]
    w = ops.composite_fwd(raw, z, d, white_bg=True)[4]
    z_all = ops.sample_fine(z, w, 128)
    raw = ops.nerf_mlp_fwd(o, d, z_all, blobs[prec_f][1], prec_f)
    if prec_f == 'bf16':
        ops.nerf_refine_last_sample(o, d, z_all, raw, gblob[1])
    return ops.composite_fwd(raw, z_all, d, white_bg=True, want_weights=False)[0]


def dilate(m, k):
    for _ in range(k):
        m = m | torch.cat((m[:, 1:], torch.zeros_like(m[:, :1])), 1) | torch.cat((torch.zeros_like(m[:, :1]), m[:, :-1]), 1)
    return m


def crit(t_min, a_lo, a_hi, dil):
    return lambda alpha, T, sig: dilate((T > t_min) & (alpha > a_lo) & (alpha < a_hi), dil)


def crit_ray(lo, hi):
    def f(alpha, T, sig):
        occu = 1 - (T[:, -1] * (1 - alpha[:, -1]))
        return ((occu > lo) & (occu < hi))[:, None].expand_as(alpha).clone()
    return f


def gated(gate, jump=None, amax=None):
    """The sample rule (visible, not saturated, +-1) on the rays whose coarse alphas show a density EDGE: adjacent samples whose
    alphas differ by >= `jump`, or one sample with alpha >= `amax`."""
    base = crit(1e-4, 1e-4, 0.9999, 1)

    def f(alpha, T, sig):
        vis = alpha * (T > 1e-4)
        g = torch.zeros(alpha.shape[0], dtype=torch.bool, device=alpha.device)
        if jump is not None:
            g |= ((vis[:, 1:-1] - vis[:, :-2]).abs().max(1)[0] >= jump)
        if amax is not None:
            g |= (vis[:, :-1].max(1)[0] >= amax)
        return base(alpha, T, sig) & g[:, None]
    return f


CRITERIA = {"T>1e-4, alpha in (1e-4, 0.9999), +-1": crit(1e-4, 1e-4, 0.9999, 1)}
for jv in (0.03, 0.05, 0.1, 0.2, 0.3):
    CRITERIA["edge: adjacent alpha jump >= %g" % jv] = gated(None, jump=jv)
for av in (0.1, 0.2, 0.3, 0.5):
    CRITERIA["edge: max alpha >= %g" % av] = gated(None, amax=av)

out = {}
for wname, nets in (("fitted", gi.trained_nerf_nets()), ("glorot_opaque", synth.nerf_nets(seed=0))):
    blobs = {p: [ops.pack_nerf_weights(*synth.nerf_layers(n), prec=p).to(dev) for n in nets] for p in ('bf16', 'fp32')}
    gblob = [ops.pack_nerf_geom_weights(*synth.nerf_layers(n), prec='fp32').to(dev) for n in nets]
    rayo, rayd = synth.camera_rays(800, 800, cam_loc=(3.2, -0.1, 2.4))
    idx = np.sort(np.random.default_rng(1).permutation(rayo.shape[0])[:N])
    o = torch.from_numpy(rayo[idx]).to(dev)
    d = ops.l2_normalize3(torch.from_numpy(rayd[idx]).to(dev), 1e-12)
    want = render(o, d, blobs, gblob, 'fp32', 'fp32')
    res = {}

    def score(rgb):
        err = (rgb - want).abs().max(1)[0]
        return dict(rays_above_3e_2=int((err > 3e-2).sum()), rays_above_2e_2=int((err > 2e-2).sum()), max_abs=float(err.max()),
                    q9999=float(torch.quantile(err, 0.9999)))
    res["bf16 coarse (shipped)"] = score(render(o, d, blobs, gblob, 'bf16', 'bf16'))
    res["fp32-class coarse"] = score(render(o, d, blobs, gblob, 'fp32', 'bf16'))
    for cname, c in CRITERIA.items():
        st = {}
        r = score(render(o, d, blobs, gblob, 'bf16', 'bf16', select=c, stats=st))
        r.update(st)
        res[cname] = r
    out[wname] = dict(rays=N, results=res)
print(json.dumps(out, indent=1))
