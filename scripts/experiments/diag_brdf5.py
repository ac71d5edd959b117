"""Which column tile / lanes differ between the per-row geometry (CT = 3) and the half-split geometry (CT = 4)?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tests.test_gpu_nerfactor import net128, pack, scene, dev
from nerfactor_amd import ops, _capi
cuda = torch.device('cuda:0')
zd, n = 3, 8
layers, out = net128(40 + zd, zd + 15, 1)
blob = pack(layers, out, _capi.IN_Z_RUSINK, 1, cuda, z_dim=zd)
rng, lxyz, _, xyz, cam, normal = scene(n, 41, 16)
z = rng.normal(size=(n, zd)).astype(np.float32)
args = (dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(z, cuda), dev(lxyz, cuda), blob)
os.environ['NFX_BRDF_VARIANT'] = '6'
os.environ['NFX_BRDF_CT'] = '3'
ref = ops.brdf_spec_fwd(*args).cpu().numpy()
os.environ['NFX_BRDF_CT'] = os.environ.get('DIAG_CT', '4')
got = ops.brdf_spec_fwd(*args).cpu().numpy()
for pt in range(n):
    front = np.flatnonzero(ref[pt] != 0)
    bad = np.flatnonzero(np.abs(got[pt] - ref[pt]) > 5e-3)
    pos = {l: i for i, l in enumerate(front.tolist())}
    qp = [pos[l] for l in bad.tolist() if l in pos]
    tiles = sorted(set((q % 128) // 32 for q in qp))
    print('point', pt, 'front rows', len(front), 'bad', len(bad), 'tiles hit', tiles, 'lanes', sorted(set(q % 32 for q in qp))[:40],
          'max', float(np.abs(got[pt] - ref[pt]).max()))
