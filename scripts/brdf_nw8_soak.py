"""Soak of the two-waves-per-SIMD form of brdf_compact_kernel (NFX_BRDF_CT=8, the default since round 3) against the
one-wave-per-SIMD form (NFX_BRDF_CT=4): REPS calls of 200 000 x 512 rows, every output compared bit for bit on the device.
Round 2 found the 8-wave form non-deterministic (a few thousand of 10^8 rows wrong per call); round 3 traced it to
v_permlane32_swap (lvis_v2.hip, brdf_compact_kernel's header).  To reproduce the fault build the library with
-DNFX_BRDF_SWAP=0 (the compiler's builtin swap) or =2 (the hand-timed asm of rounds 1-2) and point NFX_LIB_PATH at it:
    bash scripts/build_obj_variant.sh lvis_v2.hip swap0 -DNFX_BRDF_SWAP=0 -mllvm -amdgpu-mfma-vgpr-form"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import _capi, ops  # noqa: E402
from tests.test_gpu_nerfactor import net128, pack, scene, dev  # noqa: E402

cuda = torch.device('cuda:0')
zd, n = 3, int(os.environ.get('N', 200000))
reps = int(os.environ.get('REPS', 100))
layers, out = net128(40 + zd, zd + 15, 1)
blob = pack(layers, out, _capi.IN_Z_RUSINK, 1, cuda, z_dim=zd)
rng, lxyz, _, xyz, cam, normal = scene(n, 41, 16)
z = rng.normal(size=(n, zd)).astype(np.float32)
args = (dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(z, cuda), dev(lxyz, cuda), blob)


def timed(ct, k=5):
    os.environ['NFX_BRDF_CT'] = ct
    for _ in range(2):
        o = ops.brdf_spec_fwd(*args)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(k):
        o = ops.brdf_spec_fwd(*args)
    e1.record()
    torch.cuda.synchronize()
    return o, e0.elapsed_time(e1) / k


ref, t4 = timed('4')
got, t8 = timed('8')
print('4 waves x 4 tiles %.3f ms   8 waves x 2 tiles %.3f ms per call' % (t4, t8))
bad_calls, bad_rows = 0, 0
t0 = time.time()
for r in range(reps):
    got = ops.brdf_spec_fwd(*args)
    nb = int((got != ref).sum())
    if nb:
        bad_calls += 1
        bad_rows += nb
        if bad_calls <= 3:
            d = (got - ref).abs()
            print('call %d: %d rows differ, max |diff| %.3g' % (r, nb, float(d.max())))
print('%d calls x %d rows: %d calls with differences, %d rows in all (%.1f s)' % (reps, ref.numel(), bad_calls, bad_rows, time.time() - t0))
