"""brdf_spec_fwd (front-lit compaction kernel) on 200 000 points x 512 lights: time per call and the output saved /
compared bit for bit.   NFX_LIB_PATH=old.so python scripts/brdf_ab.py save;  python scripts/brdf_ab.py check"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import _capi, ops  # noqa: E402
from tests.test_gpu_nerfactor import net128, pack, scene, dev  # noqa: E402

cuda = torch.device('cuda:0')
mode = sys.argv[1]
zd, n = 3, int(os.environ.get('N', 200000))
layers, out = net128(40 + zd, zd + 15, 1)
blob = pack(layers, out, _capi.IN_Z_RUSINK, 1, cuda, z_dim=zd)
rng, lxyz, _, xyz, cam, normal = scene(n, 41, 16)
z = rng.normal(size=(n, zd)).astype(np.float32)
args = (dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(z, cuda), dev(lxyz, cuda), blob)
res = {}
for ct in os.environ.get('CTS', '4,2').split(','):
    os.environ['NFX_BRDF_CT'] = ct
    for _ in range(2):
        o = ops.brdf_spec_fwd(*args)
    torch.cuda.synchronize()
    ts = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            o = ops.brdf_spec_fwd(*args)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    res[ct] = o.cpu()
    print('%s CT=%s  %.3f ms per call (min of 3 x 5; %s)' % (mode, ct, min(ts), ' '.join('%.3f' % t for t in ts)))
path = '/tmp/brdf_ab.pt'   # (800 MB: not under gpurun_out)
if mode == 'save':
    torch.save(res, path)
else:
    ref = torch.load(path)
    for ct in res:
        same = torch.equal(res[ct], ref[ct])
        d = (res[ct] - ref[ct]).abs()
        print('CT=%s %s  (max abs diff %.3g, rows differing %d of %d)' % (
            ct, 'bit-identical' if same else 'DIFFERENT', float(d.max()), int((d > 0).sum()), d.numel()))
