"""Where the selective coarse refinement's frame time goes on the fitted view (bench: +11.8 ... 12.6 % of the frame): HIP-event
times of nfx_nerf_refine_select, nfx_nerf_sigma_refine (fp32-class density of the listed samples) and, beside them, the
fp32-class density of ALL coarse samples through the same kernel (nfx_nerf_sigma_fwd) and the bf16 coarse MLP launch."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import _capi, ops, synth  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402

dev = torch.device('cuda:0')
nets = gi.trained_nerf_nets()
blob = ops.pack_nerf_weights(*synth.nerf_layers(nets[0])).to(dev)
gblob = ops.pack_nerf_geom_weights(*synth.nerf_layers(nets[0]), prec='fp32').to(dev)
rayo, rayd = synth.camera_rays(800, 800, cam_loc=(3.2, -0.1, 2.4))
o = torch.from_numpy(rayo).to(dev)
d = ops.l2_normalize3(torch.from_numpy(rayd).to(dev), 1e-12)
z = ops.gen_z(2., 6., 64, o.shape[0], device=dev)
n, s = z.shape
raw0 = ops.nerf_mlp_fwd(o, d, z, blob)
ops.nerf_refine_last_sample(o, d, z, raw0, gblob)
margin = ops.REFINE_MARGIN_FACTOR * ops.nerf_coarse_error(o, d, z, raw0, gblob)[1]
lst = torch.empty(n * s, dtype=torch.int32, device=dev)
cnt = torch.empty(1, dtype=torch.int32, device=dev)
st = lambda: torch.cuda.current_stream().cuda_stream


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps


raw = raw0.clone()
sel = lambda: _capi.check(_capi.lib.nfx_nerf_refine_select(raw.data_ptr(), z.data_ptr(), d.data_ptr(), n, s, ops.REFINE_T_MIN, ops.REFINE_A_LO,
                                                           ops.REFINE_A_HI, margin, ops.REFINE_DILATE, lst.data_ptr(), cnt.data_ptr(), st()), 'select')
ref = lambda: _capi.check(_capi.lib.nfx_nerf_sigma_refine(o.data_ptr(), d.data_ptr(), z.data_ptr(), n, s, gblob.data_ptr(), lst.data_ptr(),
                                                          cnt.data_ptr(), raw.data_ptr(), st()), 'refine')
t_sel = timed(sel)
k = int(cnt.item())
t_ref = timed(ref)
t_all = timed(lambda: ops.nerf_sigma_fwd(o, d, z, gblob, 'fp32'), reps=2)
t_mlp = timed(lambda: ops.nerf_mlp_fwd(o, d, z, blob), reps=3)
# the same number of samples, but CONTIGUOUS (the first k): what the gather costs
zc = z.reshape(-1)[:k].reshape(-1, 1).contiguous()
oc, dc = o[:k // 1].contiguous()[:zc.shape[0]], d[:zc.shape[0]].contiguous()
t_contig = timed(lambda: ops.nerf_sigma_fwd(oc, dc, zc, gblob, 'fp32'), reps=3) if zc.shape[0] <= o.shape[0] else None
flop = 2 * (63 * 256 + 6 * 256 * 256 + 319 * 256 + 256)
print(json.dumps({"coarse_samples": n * s, "listed": k, "listed_frac": k / (n * s), "sigma_margin": margin,
                  "select_ms": t_sel, "refine_listed_ms": t_ref, "refine_listed_tflops_algorithmic": k * flop / t_ref / 1e9,
                  "fp32_class_density_of_all_coarse_samples_ms": t_all, "that_tflops_algorithmic": n * s * flop / t_all / 1e9,
                  "contiguous_samples_same_count_ms": t_contig, "bf16_coarse_mlp_ms": t_mlp}))
