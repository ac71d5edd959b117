"""How far is the bf16 kernel's density of a ray's LAST sample from the fp32-class kernel's — the quantity that decides
alpha_last = [sigma_last > 0] (nerf.py:186-191)?  800 x 800 rays of the bench view, glorot "opaque" and fitted weights,
coarse and fine network: max / quantiles of |sigma_bf16 - sigma_fp32|, how many rays change sign, how many sit within
0.06 / 0.25 of zero.  Evidence for models/nerf.py `last_sample_precision`."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import ops, synth  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402

dev = torch.device('cuda:0')
rayo, rayd = synth.camera_rays(800, 800, cam_loc=(3.2, -0.1, 2.4))
o, d = torch.from_numpy(rayo).to(dev), ops.l2_normalize3(torch.from_numpy(rayd).to(dev), 1e-12)
res = {}
for wname, nets in (('glorot_opaque', synth.nerf_nets(seed=0)), ('fitted', gi.trained_nerf_nets())):
    for i, pref in enumerate(('coarse', 'fine')):
        layers = synth.nerf_layers(nets[i])
        blob = ops.pack_nerf_weights(*layers).to(dev)
        gblob = ops.pack_nerf_geom_weights(*layers, prec='fp32').to(dev)
        z = torch.full((o.shape[0], 1), 6.0, device=dev)
        s16 = ops.nerf_mlp_fwd(o, d, z, blob)[:, 0, 3]
        s32 = ops.nerf_sigma_fwd(o, d, z, gblob, 'fp32')[:, 0]
        e = (s16 - s32).abs()
        res['%s_%s' % (wname, pref)] = dict(
            max_abs_err=float(e.max()), q999=float(torch.quantile(e, 0.999)), median=float(e.median()),
            sign_flips=int(((s16 > 0) != (s32 > 0)).sum()), rays=int(e.numel()),
            frac_abs_sigma_below_0p06=float((s32.abs() < 0.06).float().mean()),
            frac_abs_sigma_below_0p25=float((s32.abs() < 0.25).float().mean()),
            max_abs_sigma_among_flips=float(s32.abs()[(s16 > 0) != (s32 > 0)].max()) if ((s16 > 0) != (s32 > 0)).any() else 0.)
print(json.dumps(res, indent=1))
