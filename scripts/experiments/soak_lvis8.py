"""Soak test of the 8-wave light-visibility kernel (two waves per SIMD): N launches of 200 000 points x 512 lights,
every output compared bit for bit with the 4-wave kernel's."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tests.test_gpu_nerfactor import net128, pack, scene, dev
from nerfactor_amd import ops, _capi
cuda = torch.device('cuda:0')
runs = int(os.environ.get('SOAK_RUNS', '100'))
layers, out = net128(30, 90, 1)
blob = pack(layers, out, _capi.IN_XYZ_LDIR, 1, cuda)
n = 200000
rng, lxyz, _, xyz, _, _ = scene(n, 31, 16)
args = (dev(xyz, cuda), dev(lxyz, cuda), blob)
os.environ['NFX_LVIS_VARIANT'] = '4'
ref = ops.lvis_fwd(*args)
os.environ['NFX_LVIS_VARIANT'] = '8'
bad = 0
for i in range(runs):
    got = ops.lvis_fwd(*args)
    d = int((got != ref).sum())
    bad += d
    if d:
        print('run', i, 'rows differing', d, flush=True)
print('lvis variant 8 vs 4: %d launches x %d rows, rows differing in total: %d' % (runs, n * lxyz.shape[0], bad))
