"""Second probe of the selective fp32-class coarse refinement (see coarse_refine_probe.py): the first one showed that NO criterion
on the bf16 pass's alpha / transmittance alone separates "matters" from "does not" across weight sets (fitted scene: the samples
that decide are at the density's zero crossing, where the bf16 kernel may say alpha = 0; glorot fog: every sample is partially
transparent and none matters).  What separates them is the SIZE of the bf16 error, which scales with the activations:
    E_i = sum_k |w_sigma[k]| h_k(x_i)      (h = the encoder's output, >= 0 after ReLU: E is one more linear row next to sigma)
    delta_i = c * 2^-9 * E_i               (operand rounding of the last layer, c for what the eight layers before it add)
A sample is re-evaluated when its error bar can move the weights:  T_up_i * (alpha(sigma_i + delta_i) - alpha(sigma_i - delta_i)) > theta.
Here E comes from a torch fp32 forward (calibration only); reports the distribution of |sigma_bf16 - sigma_fp32class| / (2^-9 E)
and, per (c, theta), the selected fraction and the rays above 3e-2 / 2e-2 against the fp32-class render."""
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


def embed(x, n):
    parts = [x]
    for k in range(n):
        parts += [torch.sin(x * float(2 ** k)), torch.cos(x * float(2 ** k))]
    return torch.cat(parts, -1)


def sigma_and_E(o, d, z, net):
    """fp32 torch forward of the encoder: (sigma, E = |w_sigma| . h) per sample."""
    ks, bs = synth.nerf_layers(net)
    ks = [torch.from_numpy(np.asarray(k, np.float32)).to(dev) for k in ks]
    bs = [torch.from_numpy(np.asarray(b, np.float32)).to(dev) for b in bs]
    sig, E = [], []
    for i in range(0, o.shape[0], 8192):
        pts = (o[i:i + 8192, None] + d[i:i + 8192, None] * z[i:i + 8192, :, None]).reshape(-1, 3)
        pe = embed(pts, 10)
        h = pe
        for l in range(8):
            h = torch.relu(h @ ks[l] + bs[l])
            if l == 4:
                h = torch.cat((h, pe), -1)
        sig.append((h @ ks[8] + bs[8]).reshape(-1, z.shape[1]))
        E.append((h @ ks[8].abs()).reshape(-1, z.shape[1]))
    return torch.cat(sig), torch.cat(E)


def finish(o, d, z, raw, blobs, gblob):
    w = ops.composite_fwd(raw, z, d, white_bg=True)[4]
    z_all = ops.sample_fine(z, w, 128)
    raw_f = ops.nerf_mlp_fwd(o, d, z_all, blobs['bf16'][1], 'bf16')
    ops.nerf_refine_last_sample(o, d, z_all, raw_f, gblob[1])
    return ops.composite_fwd(raw_f, z_all, d, white_bg=True, want_weights=False)[0]


out = {}
for wname, nets in (("fitted", gi.trained_nerf_nets()), ("glorot_opaque", synth.nerf_nets(seed=0))):
    blobs = {p: [ops.pack_nerf_weights(*synth.nerf_layers(n), prec=p).to(dev) for n in nets] for p in ('bf16', 'fp32')}
    gblob = [ops.pack_nerf_geom_weights(*synth.nerf_layers(n), prec='fp32').to(dev) for n in nets]
    rayo, rayd = synth.camera_rays(800, 800, cam_loc=(3.2, -0.1, 2.4))
    idx = np.sort(np.random.default_rng(1).permutation(rayo.shape[0])[:N])
    o = torch.from_numpy(rayo[idx]).to(dev)
    d = ops.l2_normalize3(torch.from_numpy(rayd[idx]).to(dev), 1e-12)
    z = ops.gen_z(2., 6., 64, o.shape[0], device=dev)
    # reference: fp32-class coarse + fp32-class fine
    raw32 = ops.nerf_mlp_fwd(o, d, z, blobs['fp32'][0], 'fp32')
    w = ops.composite_fwd(raw32, z, d, white_bg=True)[4]
    z_all = ops.sample_fine(z, w, 128)
    want = ops.composite_fwd(ops.nerf_mlp_fwd(o, d, z_all, blobs['fp32'][1], 'fp32'), z_all, d, white_bg=True, want_weights=False)[0]
    raw16 = ops.nerf_mlp_fwd(o, d, z, blobs['bf16'][0], 'bf16')
    ops.nerf_refine_last_sample(o, d, z, raw16, gblob[0])
    s32 = ops.nerf_sigma_fwd(o, d, z, gblob[0], 'fp32')               # fp32-class density of every coarse sample
    _, E = sigma_and_E(o, d, z, nets[0])
    dsig = (raw16[..., 3] - s32).abs()
    ratio = (dsig / (2. ** -9 * E.clamp_min(1e-12)))[:, :-1].reshape(-1)
    sub = ratio[torch.randperm(ratio.numel(), device=dev)[:4000000]]
    res = {"abs_dsigma_over_2^-9_E quantiles (50, 90, 99, 99.9, 99.99 %, max)":
           [float(torch.quantile(sub, q)) for q in (0.5, 0.9, 0.99, 0.999, 0.9999)] + [float(ratio.max())],
           "abs_dsigma quantiles (50, 99, 99.99 %, max)": [float(torch.quantile(dsig.reshape(-1)[:4000000], q)) for q in (0.5, 0.99, 0.9999)] + [float(dsig.max())],
           "E quantiles (50, 99, max)": [float(torch.quantile(E.reshape(-1)[:4000000], q)) for q in (0.5, 0.99)] + [float(E.max())]}

    def score(rgb):
        err = (rgb - want).abs().max(1)[0]
        return dict(rays_above_3e_2=int((err > 3e-2).sum()), rays_above_2e_2=int((err > 2e-2).sum()), max_abs=float(err.max()))
    res["bf16 coarse (shipped)"] = score(finish(o, d, z, raw16.clone(), blobs, gblob))
    dist = torch.cat((z[:, 1:] - z[:, :-1], torch.full_like(z[:, :1], 1e10)), 1) * d.norm(dim=1, keepdim=True)
    sig16 = raw16[..., 3]
    for c in (1., 2., 4., 8.):
        delta = c * 2. ** -9 * E
        a_lo = 1 - torch.exp(-torch.relu(sig16 - delta) * dist)
        a_hi = 1 - torch.exp(-torch.relu(sig16 + delta) * dist)
        T_up = torch.cumprod(torch.cat((torch.ones_like(a_lo[:, :1]), 1 - a_lo[:, :-1] + 1e-6), 1), 1).clamp(max=1.)
        gain = T_up * (a_hi - a_lo)
        for theta in (2e-3, 5e-3, 1e-2, 2e-2, 4e-2):
            m = gain > theta
            m[:, -1] = False
            raw = raw16.clone()
            raw[..., 3] = torch.where(m, s32, raw[..., 3])
            r = score(finish(o, d, z, raw, blobs, gblob))
            r.update(selected_frac=float(m.float().mean()), rays_touched_frac=float(m.any(1).float().mean()))
            res["c=%g theta=%g" % (c, theta)] = r
    out[wname] = dict(rays=N, results=res)
print(json.dumps(out, indent=1))
