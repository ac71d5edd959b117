"""What is left after the selective coarse refinement (27 of 640 000 rays of the fitted view above 3e-2 of the fp32-class
render): is it the coarse pass's remaining bf16 samples, or the bf16 FINE pass?  Renders the whole view with
(a) coarse fp32-class + fine bf16, (b) coarse refined + fine fp32-class, (c) wider selection rules, against all-fp32-class."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import ops, synth  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402

dev = torch.device('cuda:0')
nets = gi.trained_nerf_nets()
blobs = {p: [ops.pack_nerf_weights(*synth.nerf_layers(n), prec=p).to(dev) for n in nets] for p in ('bf16', 'fp32')}
gblob = [ops.pack_nerf_geom_weights(*synth.nerf_layers(n), prec='fp32').to(dev) for n in nets]
rayo, rayd = synth.camera_rays(800, 800, cam_loc=(1.9, -2.8, 2.1))
o = torch.from_numpy(rayo).to(dev)
d = ops.l2_normalize3(torch.from_numpy(rayd).to(dev), 1e-12)
z = ops.gen_z(2., 6., 64, o.shape[0], device=dev)


def render(pc, pf, sel=None):
    raw = ops.nerf_mlp_fwd(o, d, z, blobs[pc][0], pc)
    frac = None
    if pc == 'bf16':
        ops.nerf_refine_last_sample(o, d, z, raw, gblob[0])
        if sel is not None:
            _, cnt = ops.nerf_refine_coarse(o, d, z, raw, gblob[0], want_count=True, **sel)
            frac = float(cnt.item()) / z.numel()
    w = ops.composite_fwd(raw, z, d, white_bg=True)[4]
    z_all = ops.sample_fine(z, w, 128)
    raw = ops.nerf_mlp_fwd(o, d, z_all, blobs[pf][1], pf)
    if pf == 'bf16':
        ops.nerf_refine_last_sample(o, d, z_all, raw, gblob[1])
    return ops.composite_fwd(raw, z_all, d, white_bg=True, want_weights=False)[0], frac


want, _ = render('fp32', 'fp32')
out = {}
variants = [("coarse fp32-class, fine bf16", ('fp32', 'bf16', None)),
            ("coarse bf16, fine bf16", ('bf16', 'bf16', None)),
            ("default rule (visible, unsaturated, +-1)", ('bf16', 'bf16', {}))]
for m in (0.1, 0.3, 1.0, 3.0):
    variants.append(("ONLY sign-undecided |sigma| < %g (T > 1e-4)" % m, ('bf16', 'bf16', dict(a_lo=2., a_hi=-1., dilate=0, sigma_margin=m))))
    variants.append(("ONLY sign-undecided |sigma| < %g (T > 1e-4) +-1" % m, ('bf16', 'bf16', dict(a_lo=2., a_hi=-1., dilate=1, sigma_margin=m))))
for t_min, a_lo, a_hi in ((1e-4, 1e-4, 0.9999), (1e-3, 1e-3, 0.999), (1e-2, 1e-2, 0.99), (1e-2, 5e-2, 0.95), (5e-2, 0.1, 0.9)):
    for dil in (0, 1):
        variants.append(("T > %g, alpha in (%g, %g), +-%d, |sigma| < 0.3" % (t_min, a_lo, a_hi, dil),
                         ('bf16', 'bf16', dict(t_min=t_min, a_lo=a_lo, a_hi=a_hi, dilate=dil, sigma_margin=0.3))))
for name, args in variants:
    rgb, frac = render(*args)
    err = (rgb - want).abs().max(1)[0]
    out[name] = dict(rays_above_3e_2=int((err > 3e-2).sum()), rays_above_2e_2=int((err > 2e-2).sum()), max_abs=float(err.max()), refined_frac=frac)
print(json.dumps(out, indent=1))
