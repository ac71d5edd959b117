"""Is the 8-wave light-visibility kernel (variant 8, two waves per SIMD) bit-identical to variant 4 at scale, run
after run?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tests.test_gpu_nerfactor import net128, pack, scene, dev
from nerfactor_amd import ops, _capi
cuda = torch.device('cuda:0')
layers, out = net128(30, 90, 1)
blob = pack(layers, out, _capi.IN_XYZ_LDIR, 1, cuda)
for n in (1500, 200000):
    rng, lxyz, _, xyz, _, _ = scene(n, 31, 16)
    args = (dev(xyz, cuda), dev(lxyz, cuda), blob)
    os.environ['NFX_LVIS_VARIANT'] = '4'
    ref = ops.lvis_fwd(*args)
    same4 = all(bool(torch.equal(ref, ops.lvis_fwd(*args))) for _ in range(3))
    os.environ['NFX_LVIS_VARIANT'] = '8'
    runs = [ops.lvis_fwd(*args) for _ in range(8)]
    print('lvis n', n, 'variant 4 repeatable', same4, '| variant 8 == variant 4 in runs:',
          [bool(torch.equal(ref, r)) for r in runs], 'max diff', max(float((r - ref).abs().max()) for r in runs),
          'rows differing', [int((r != ref).sum()) for r in runs], flush=True)
