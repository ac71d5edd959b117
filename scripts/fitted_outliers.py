"""Where do the rays of the FITTED NeRF that miss max-abs 3e-2 come from (bench.py parity_fitted_weights: 4 of 2048, the
same 4 with and without the fp32-class last sample)?  The same 2048 rays of the bench view rendered with the coarse and
the fine pass each in bf16 or fp32-class (hi / lo operand pairs), against the fp32 CPU oracle: rays above 3e-2 per
combination, and what distinguishes them (coarse occupancy, largest shift of a fine sample)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import ops, synth  # noqa: E402
from oracle import torch_ref  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402

dev = torch.device('cuda:0')
nets = gi.trained_nerf_nets()
rayo, rayd = synth.camera_rays(800, 800, cam_loc=(3.2, -0.1, 2.4))
idx = np.sort(np.random.default_rng(1).permutation(rayo.shape[0])[:2048])
o_h, d_h = rayo[idx], rayd[idx]
torch.set_num_threads(os.cpu_count() or 8)
with torch.no_grad():
    ref = torch_ref.render_rays(torch.from_numpy(o_h), torch.from_numpy(d_h), *[torch_ref.to_torch_net(x) for x in nets])
want, occu_c = ref[1]['rgb'].numpy(), ref[0]['occu'].numpy()
blob = {p: [ops.pack_nerf_weights(*synth.nerf_layers(n), prec=p).to(dev) for n in nets] for p in ('bf16', 'fp32')}
gblob = [ops.pack_nerf_geom_weights(*synth.nerf_layers(n), prec='fp32').to(dev) for n in nets]
o, d = torch.from_numpy(o_h).to(dev), ops.l2_normalize3(torch.from_numpy(d_h).to(dev), 1e-12)
out = {}
zs = {}
for pc in ('bf16', 'fp32'):
    for pf in ('bf16', 'fp32'):
        z = ops.gen_z(2., 6., 64, o.shape[0], device=dev)
        raw = ops.nerf_mlp_fwd(o, d, z, blob[pc][0], pc)
        if pc == 'bf16':
            ops.nerf_refine_last_sample(o, d, z, raw, gblob[0])
        w = ops.composite_fwd(raw, z, d, white_bg=True)[4]
        z_all = ops.sample_fine(z, w, 128)
        raw = ops.nerf_mlp_fwd(o, d, z_all, blob[pf][1], pf)
        if pf == 'bf16':
            ops.nerf_refine_last_sample(o, d, z_all, raw, gblob[1])
        rgb = ops.composite_fwd(raw, z_all, d, white_bg=True, want_weights=False)[0].cpu().numpy()
        err = np.abs(rgb - want).max(1)
        bad = np.nonzero(err > 3e-2)[0]
        zs[pc] = z_all.cpu().numpy()
        out['coarse_%s__fine_%s' % (pc, pf)] = dict(
            rays_above_3e_2=int(bad.size), max_abs=float(err.max()), q999=float(np.quantile(err, 0.999)),
            bad_rays=[dict(ray=int(idx[b]), err=float(err[b]), coarse_occupancy=float(occu_c[b])) for b in bad[:8]])
dz = np.abs(zs['bf16'] - zs['fp32'])
out['fine_sample_shift_bf16_vs_fp32_coarse'] = dict(max=float(dz.max()), q999=float(np.quantile(dz.max(1), 0.999)),
                                                      rays_with_shift_above_0p03=int((dz.max(1) > 0.03).sum()))
out['silhouette_rays_coarse_occupancy_in_0p02_0p98'] = int(((occu_c > 0.02) & (occu_c < 0.98)).sum())
print(json.dumps(out, indent=1))
