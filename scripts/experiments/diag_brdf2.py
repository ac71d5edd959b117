"""Determinism / correctness / time of the compaction kernel instantiations of whatever library NFX_LIB_PATH names."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tests.test_gpu_nerfactor import net128, pack, scene, dev
from nerfactor_amd import ops, _capi
cuda = torch.device('cuda:0')
print('lib', _capi.LIB_PATH)
for zd, n in ((3, 1500), (3, 200000)):
    layers, out = net128(40 + zd, zd + 15, 1)
    blob = pack(layers, out, _capi.IN_Z_RUSINK, 1, cuda, z_dim=zd)
    rng, lxyz, _, xyz, cam, normal = scene(n, 41, 16)
    z = rng.normal(size=(n, zd)).astype(np.float32)
    args = (dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(z, cuda), dev(lxyz, cuda), blob)
    os.environ['NFX_BRDF_VARIANT'] = '3'; os.environ['NFX_BRDF_CT'] = '4'
    ref = ops.brdf_spec_fwd(*args)
    for var, ct in (('6', '8'), ('5', '8'), ('6', '2'), ('5', '2'), ('6', '4')):
        os.environ['NFX_BRDF_VARIANT'] = var; os.environ['NFX_BRDF_CT'] = ct
        runs = [ops.brdf_spec_fwd(*args) for _ in range(4)]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(3): ops.brdf_spec_fwd(*args)
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 3 * 1e3
        det = all(bool(torch.equal(runs[0], r)) for r in runs[1:])
        d = (runs[0] - ref).abs()
        print('n %6d variant %s ct %2s deterministic %-5s max|diff vs dense| %.4f rows>1e-2 %d  %.2f ms' % (
            n, var, ct, det, float(d.max()), int((d > 1e-2).sum()), ms), flush=True)
