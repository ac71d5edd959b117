"""8-wave vs 4-wave compaction kernels, same CT = 2: bit-equal outputs?  (with -DNFX_LV3_DBG_INPUTS the output is the
sum of each row's input slots, i.e. the network is out of the picture)"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tests.test_gpu_nerfactor import net128, pack, scene, dev
from nerfactor_amd import ops, _capi
cuda = torch.device('cuda:0')
print('lib', _capi.LIB_PATH)
zd = 3
for n in (1500, 200000):
    layers, out = net128(40 + zd, zd + 15, 1)
    blob = pack(layers, out, _capi.IN_Z_RUSINK, 1, cuda, z_dim=zd)
    rng, lxyz, _, xyz, cam, normal = scene(n, 41, 16)
    z = rng.normal(size=(n, zd)).astype(np.float32)
    args = (dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(z, cuda), dev(lxyz, cuda), blob)
    for var in ('6', '5'):
        os.environ['NFX_BRDF_VARIANT'] = var
        os.environ['NFX_BRDF_CT'] = '2'
        ref = ops.brdf_spec_fwd(*args)
        os.environ['NFX_BRDF_CT'] = '8'
        runs = [ops.brdf_spec_fwd(*args) for _ in range(6)]
        print('n', n, 'variant', var, '8-wave == 4-wave:', [bool(torch.equal(ref, r)) for r in runs],
              'rows differing', [int((r != ref).sum()) for r in runs], flush=True)
