"""Which rows does the 8-wave compaction kernel get wrong, and what do they hold?"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from tests.test_gpu_nerfactor import net128, pack, scene, dev
from nerfactor_amd import ops, _capi
cuda = torch.device('cuda:0')
zd, n = 3, 1500
layers, out = net128(40 + zd, zd + 15, 1)
blob = pack(layers, out, _capi.IN_Z_RUSINK, 1, cuda, z_dim=zd)
rng, lxyz, _, xyz, cam, normal = scene(n, 41, 16)
z = rng.normal(size=(n, zd)).astype(np.float32)
args = (dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(z, cuda), dev(lxyz, cuda), blob)
os.environ['NFX_BRDF_VARIANT'] = '6'; os.environ['NFX_BRDF_CT'] = '4'
ref = ops.brdf_spec_fwd(*args).cpu().numpy()
L = ref.shape[1]
os.environ['NFX_BRDF_CT'] = os.environ.get('DIAG_CT', '8')
for run in range(6):
    junk = torch.full((n, L), float('nan'), device=cuda); del junk
    got = ops.brdf_spec_fwd(*args).cpu().numpy()
    bad = np.argwhere(~(np.abs(got - ref) <= 1e-3))
    print('run', run, 'bad rows', len(bad), 'NaN', int(np.isnan(got).sum()))
    pts = sorted(set(bad[:, 0].tolist()))
    for pt in pts[:6]:
        ls = bad[bad[:, 0] == pt][:, 1]
        front = np.flatnonzero(ref[pt] != 0)          # the point's queued rows, in queue order
        rank = {l: i for i, l in enumerate(front.tolist())}
        print('  point', pt, '(wave %d of its block, block %d)' % (pt % 8, (pt // 8) % 256), 'front rows', len(front),
              'bad lights', ls.tolist())
        print('     queue positions of the bad rows', [rank.get(int(l), -1) for l in ls])
        for l in ls[:4]:
            g, w = got[pt, l], ref[pt, l]
            # does the wrong value appear elsewhere in the reference output?
            hit = np.argwhere(np.abs(ref - g) < 1e-6)
            print('     l', int(l), 'got', g, 'want', w, 'same value in ref at', hit[:3].tolist())
