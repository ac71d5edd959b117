"""Per-tile cycle stamps of one wave of the resident light-visibility kernel (diagnostic build:
NFX_EXTRA_DEFS=-DNFX_LV2_TIMING python -m nerfactor_amd.build --out nerfactor_amd/libnfx_t.so; NFX_LIB_PATH=...)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import _capi, ops  # noqa: E402
from oracle import nerfactor_ref as R  # noqa: E402

dev = torch.device('cuda:0')
rng = np.random.default_rng(0)
layers, fan = [], 90
for i in range(4):
    layers.append((rng.normal(size=(fan, 128)).astype(np.float32) * 0.1, np.zeros(128, np.float32)))
    fan = 128 + (90 if i == 2 else 0)
out = (rng.normal(size=(128, 1)).astype(np.float32) * 0.1, np.zeros(1, np.float32))
ks = [k for k, _ in layers] + [out[0]]
bs = [b for _, b in layers] + [out[1]]
blob = ops.pack_mlp128_weights(ks, bs, _capi.IN_XYZ_LDIR, 1).to(dev)
n = 200000
xyz = torch.rand(n, 3, device=dev) * 2 - 1
lxyz, _ = R.gen_light_xyz(16, 32)
lxyz = torch.from_numpy(lxyz.reshape(-1, 3).astype(np.float32)).to(dev)
for _ in range(2):
    ops.lvis_fwd(xyz, lxyz, blob)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
_capi.lib.nfx_debug_lv2_times.argtypes = [ctypes.c_void_p]
assert _capi.lib.nfx_debug_lv2_times(buf) == 0
t = np.array(buf[:18], dtype=np.int64)
d = np.diff(t)
names = ['L0'] * 4 + ['L1'] * 4 + ['L2'] * 4 + ['L3'] * 4 + ['out']
mf = [8] * 4 + [32] * 8 + [40] * 4 + [32]
print("tile  layer  cycles  mfma_cycles(32/MFMA at CT=4)")
for i, (c, nm, m) in enumerate(zip(d, names, mf)):
    print("%3d   %-4s  %6d  %5d" % (i, nm, c, m * 32))
print("total %d cycles for %d MFMA cycles (%.1f %%)" % (d.sum(), sum(mf) * 32, 100. * sum(mf) * 32 / d.sum()))

for base, name in ((20, 'tile 8 (first of L2)'), (32, 'tile 9')):
    tt = np.array(buf[base:base + 8], dtype=np.int64)
    print(name, 'k-step stamps (after the MFMAs of step s are issued), deltas:', np.diff(tt).tolist())
