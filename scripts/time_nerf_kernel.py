"""Wall time of the NeRF MLP kernel alone (HIP events), for A/B builds selected with NFX_LIB_PATH.
    NFX_LIB_PATH=... python scripts/time_nerf_kernel.py [rays] [samples]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import ops, synth  # noqa: E402

dev = torch.device('cuda:0')
blob = ops.pack_nerf_weights(*synth.nerf_layers(synth.nerf_nets(seed=0)[0])).to(dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 640000
s = int(sys.argv[2]) if len(sys.argv) > 2 else 192
o = torch.randn(n, 3, device=dev)
d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
z = torch.sort(torch.rand(n, s, device=dev) * 4 + 2, dim=1)[0]
for _ in range(2):
    ops.nerf_mlp_fwd(o, d, z, blob)
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    ops.nerf_mlp_fwd(o, d, z, blob)
    e1.record()
    torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
ms = min(ts)
print("%s: %.2f ms (min of 5; all %s) = %.1f TFLOP/s" % (os.environ.get('NFX_LIB_PATH', 'libnfx.so').split('/')[-1], ms,
                                                        ' '.join('%.2f' % t for t in ts), n * s * 1186816 / ms / 1e9))
