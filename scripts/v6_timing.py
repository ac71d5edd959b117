"""Per-tile cycle stamps of the four waves of one workgroup of the variant 6 / 7 / 8 NeRF MLP kernel (diagnostic build:
NFX_EXTRA_DEFS=-DNFX_V6_TIMING python -m nerfactor_amd.build --out nerfactor_amd/libnfx_t.so;
NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_t.so [NFX_NERF_VARIANT=7] python scripts/v6_timing.py).  The stamps cost a
few per cent (s_memtime shares lgkmcnt with the LDS reads)."""
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
os.environ.setdefault('NFX_NERF_VARIANT', '7')
for _ in range(2):
    ops.nerf_mlp_fwd(o, d, z, blob)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 512)()
_capi.lib.nfx_debug_v6_times.argtypes = [ctypes.c_void_p]
assert _capi.lib.nfx_debug_v6_times(buf) == 0
t = np.array(buf[:], dtype=np.int64).reshape(4, 128)[:, :78]
dt = np.diff(t, axis=1)                                    # [wave, tile]
names = ['L%d' % (i // 8) for i in range(64)] + ['bott'] * 8 + ['sig'] + ['rgb0'] * 4 + ['rgb1']
ks = np.array([4] * 8 + [16] * 32 + [20] * 8 + [16] * 16 + [16] * 9 + [18] * 4 + [8])
print("tile layer  cycles(w0..w3)            mfma_cycles  skew_at_entry")
for i in range(77):
    print("%3d  %-5s %s  %5d  %5d" % (i, names[i], ' '.join('%6d' % c for c in dt[:, i]), ks[i] * 64,
                                      int(t[:, i].max() - t[:, i].min())))
w0 = dt[0]
print("first tiles of layers (wave 0):", [int(w0[i]) for i in range(0, 72, 8)], "steady median:", int(np.median(w0[8:64])))
print("total (77 tiles, wave 0) %d cycles, MFMA cycles %d (%.1f %%)" % (w0.sum(), ks[:77].sum() * 64,
                                                                       100. * ks[:77].sum() * 64 / w0.sum()))
