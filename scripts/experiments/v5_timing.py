"""Per-tile cycle stamps of one wave of the variant-5 NeRF MLP kernel (diagnostic build:
NFX_EXTRA_DEFS=-DNFX_V5_TIMING python -m nerfactor_amd.build --out nerfactor_amd/libnfx_t.so; NFX_LIB_PATH=...)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import _capi, ops  # noqa: E402
from tests import common  # noqa: E402

dev = torch.device('cuda:0')
blob = ops.pack_nerf_weights(*common.nerf_layers(common.nerf_nets(seed=0)[0])).to(dev)
n, s = 200000, 192
o = torch.randn(n, 3, device=dev)
d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
z = torch.sort(torch.rand(n, s, device=dev) * 4 + 2, dim=1)[0]
os.environ['NFX_NERF_VARIANT'] = '5'
for _ in range(2):
    ops.nerf_mlp_fwd(o, d, z, blob)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 128)()
_capi.lib.nfx_debug_v5_times.argtypes = [ctypes.c_void_p]
assert _capi.lib.nfx_debug_v5_times(buf) == 0
t = np.array(buf[:78], dtype=np.int64)
dt = np.diff(t)
names = ['L%d' % (i // 8) for i in range(64)] + ['bott'] * 8 + ['sig'] + ['rgb0'] * 4 + ['rgb1']
ks = [4] * 8 + [16] * 32 + [20] * 8 + [16] * 16 + [16] * 9 + [18] * 4 + [8]
print("tile layer cycles mfma_cycles")
for i, c in enumerate(dt):
    print("%3d  %-5s %6d  %5d" % (i, names[i], c, ks[i] * 2 * 32))
print("first tiles of layers:", [int(dt[i]) for i in range(0, 72, 8)], "steady median:", int(np.median(dt[8:64])))
print("total (77 tiles) %d cycles, MFMA cycles %d (%.1f %%)" % (dt.sum(), sum(ks[:77]) * 64, 100. * sum(ks[:77]) * 64 / dt.sum()))
