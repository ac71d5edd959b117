"""Where one wave of brdf_compact_kernel spends its cycles: queue fill | row gathers + Rusinkiewicz geometry + operand
build | the 17 MFMA tiles (diagnostic build: NFX_EXTRA_DEFS=-DNFX_LV2_TIMING python -m nerfactor_amd.build --out
nerfactor_amd/libnfx_t.so; NFX_LIB_PATH=nerfactor_amd/libnfx_t.so python scripts/brdf_phases.py)."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import _capi, ops  # noqa: E402
from tests.test_gpu_nerfactor import net128, pack, scene, dev  # noqa: E402

cuda = torch.device('cuda:0')
zd, n = 3, int(os.environ.get('N', 200000))
layers, out = net128(40 + zd, zd + 15, 1)
blob = pack(layers, out, _capi.IN_Z_RUSINK, 1, cuda, z_dim=zd)
rng, lxyz, _, xyz, cam, normal = scene(n, 41, 16)
z = rng.normal(size=(n, zd)).astype(np.float32)
args = (dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(z, cuda), dev(lxyz, cuda), blob)
for ct in (os.environ.get('CTS', '4,2')).split(','):
    os.environ['NFX_BRDF_CT'] = ct
    for _ in range(2):
        ops.brdf_spec_fwd(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.brdf_spec_fwd(*args)
    e1.record()
    torch.cuda.synchronize()
    buf = (ctypes.c_ulonglong * 64)()
    _capi.lib.nfx_debug_lv2_times.argtypes = [ctypes.c_void_p]
    assert _capi.lib.nfx_debug_lv2_times(buf) == 0
    fill, geo, net, passes = (int(buf[i]) for i in (48, 49, 50, 51))
    tot = fill + geo + net
    mf = (4 * 2 + 8 * 8 + 4 * 10 + 8) * int(ct) * 32
    print('CT=%s  %.3f ms per call (%d x %d rows)   wave 0 of block 7: %d passes; cycles per pass: fill %.0f (%.1f %%)  '
          'gather+geometry %.0f (%.1f %%)  tiles %.0f (%.1f %%; MFMA issue alone %d)' % (
              ct, e0.elapsed_time(e1) / 5, n, lxyz.shape[0], passes, fill / passes, 100. * fill / tot, geo / passes,
              100. * geo / tot, net / passes, 100. * net / tot, mf))
    t = np.array(buf[:18], dtype=np.int64)
    d = np.diff(t)
    ks = [2] * 4 + [8] * 8 + [10] * 4 + [8]
    print('   pass 40, per tile: cycles / MFMA issue cycles: ' + '  '.join('%d/%d' % (c, k * int(ct) * 32) for c, k in zip(d, ks)))
