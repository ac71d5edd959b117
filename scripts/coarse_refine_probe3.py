"""Third probe of the selective coarse refinement: RAY-level rules from the conditioning of the inverse-CDF sampler.  Probe 2:
the bf16 density errors are small everywhere (fitted scene: <= 0.3, i.e. <= 2 % of alpha per sample) — the rays above 3e-2
are rays where the sampler is ILL-CONDITIONED: a fine sample placed in a bin of tiny probability moves by dz = d cdf / pdf_bin
bins, and the fine network has structure there.  Rules (all from the bf16 coarse pass alone):
  R1(tau): some fine sample lies in a bin with pdf_bin < tau;      R2(tau): some bin has 0 < pdf_bin < tau;
  R4(tau, k): at least k fine samples lie in bins with pdf_bin < tau.
Marked rays take the fp32-class density for ALL their coarse samples.  Reports marked fraction and rays above 3e-2 / 2e-2."""
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
    raw32 = ops.nerf_mlp_fwd(o, d, z, blobs['fp32'][0], 'fp32')
    w32 = ops.composite_fwd(raw32, z, d, white_bg=True)[4]
    z_all = ops.sample_fine(z, w32, 128)
    want = ops.composite_fwd(ops.nerf_mlp_fwd(o, d, z_all, blobs['fp32'][1], 'fp32'), z_all, d, white_bg=True, want_weights=False)[0]
    raw16 = ops.nerf_mlp_fwd(o, d, z, blobs['bf16'][0], 'bf16')
    ops.nerf_refine_last_sample(o, d, z, raw16, gblob[0])
    s32 = ops.nerf_sigma_fwd(o, d, z, gblob[0], 'fp32')
    w16 = ops.composite_fwd(raw16, z, d, white_bg=True)[4]
    # the sampler's bins as util/math.py:71-94 builds them
    pdf = w16[:, 1:-1] / (w16[:, 1:-1].sum(1, keepdim=True) + 1e-5)                # [n, 62]
    cdf = torch.cat((torch.zeros_like(pdf[:, :1]), torch.cumsum(pdf, 1)), 1)       # [n, 63]
    u = torch.linspace(0, 1, 128, device=dev)[None].expand(o.shape[0], -1).contiguous()
    ind = torch.searchsorted(cdf, u, right=True)
    below, above = (ind - 1).clamp(min=0), ind.clamp(max=62)
    pbin = torch.gather(cdf, 1, above) - torch.gather(cdf, 1, below)                # probability of the bin each fine sample lies in
    pbin = torch.where(pbin < 1e-5, torch.ones_like(pbin), pbin)                    # (den < 1e-5 -> 1: a degenerate bin puts the sample on its edge)
    err0 = (finish(o, d, z, raw16.clone(), blobs, gblob) - want).abs().max(1)[0]
    bad0 = err0 > 3e-2
    res = {"bf16 coarse (shipped)": dict(rays_above_3e_2=int(bad0.sum()), max_abs=float(err0.max())),
           "min pdf_bin over the fine samples: quantiles (1, 10, 50 %) over BAD rays": [float(torch.quantile(pbin.min(1)[0][bad0], q)) for q in (0.01, 0.1, 0.5)] if bad0.any() else None,
           "the same over ALL rays": [float(torch.quantile(pbin.min(1)[0], q)) for q in (0.01, 0.1, 0.5)]}

    def score(mark):
        raw = raw16.clone()
        raw[..., 3] = torch.where(mark[:, None], s32, raw[..., 3])
        err = (finish(o, d, z, raw, blobs, gblob) - want).abs().max(1)[0]
        return dict(marked_frac=float(mark.float().mean()), rays_above_3e_2=int((err > 3e-2).sum()), rays_above_2e_2=int((err > 2e-2).sum()),
                    max_abs=float(err.max()), bad_rays_marked="%d of %d" % (int((mark & bad0).sum()), int(bad0.sum())))
    for tau in (1e-4, 3e-4, 1e-3, 3e-3, 6e-3, 1e-2):
        res["R1 tau=%g" % tau] = score((pbin < tau).any(1))
        res["R2 tau=%g" % tau] = score(((pdf > 0) & (pdf < tau)).any(1))
    for tau, k in ((3e-3, 2), (3e-3, 4), (6e-3, 4), (1e-2, 4), (1e-2, 8)):
        res["R4 tau=%g k=%d" % (tau, k)] = score((pbin < tau).sum(1) >= k)
    out[wname] = dict(rays=N, results=res)
print(json.dumps(out, indent=1))
