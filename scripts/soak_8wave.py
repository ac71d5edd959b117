"""Soak of every two-waves-per-SIMD kernel of lvis_v2.hip against its one-wave-per-SIMD form, bit for bit on the device
(VERDICT r03 #1): REPS launches at each size, every output compared; for a launch that differs the number of rows, the
position of the differing elements modulo 64 / 16 (a lane-group pattern) and the largest difference are printed.

    resident128_kernel<2, 0, 8>   light visibility, the DEFAULT          (option lvis_variant 8 against 4)
    brdf_compact_kernel<2, 1, 8>  learned BRDF, closed-form angles, OPT-IN (brdf_variant 6, brdf_ct 8 against 4)
    brdf_compact_kernel<2, 0, 8>  learned BRDF, per-row geometry — built only with -DNFX_EXPERIMENT_BUILD
                                  (brdf_variant 5, brdf_ct 8 against 4); r03: failed bit identity on a fresh MI355X

    python scripts/soak_8wave.py                       # the two kernels of the product library
    NFX_LIB_PATH=.../libnfx_xp.so python scripts/soak_8wave.py --geo0    # + <2, 0, 8> from an experiment build:
        NFX_EXTRA_DEFS=-DNFX_EXPERIMENT_BUILD python -m nerfactor_amd.build --out nerfactor_amd/libnfx_xp.so
Environment: REPS (default 40), SIZES (default "333,200000")."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import _capi, ops  # noqa: E402
from tests.test_gpu_nerfactor import net128, pack, scene, dev  # noqa: E402

cuda = torch.device('cuda:0')
reps = int(os.environ.get('REPS', 40))
sizes = [int(s) for s in os.environ.get('SIZES', '333,200000').split(',')]
zd = 3
summary = []


def soak(label, call, ref_opts, test_opts, n):
    for k, v in ref_opts.items():
        _capi.set_option(k, v)
    ref = call()
    ref2 = call()
    assert torch.equal(ref, ref2), label + ': the one-wave-per-SIMD reference itself is not deterministic'
    for k, v in test_opts.items():
        _capi.set_option(k, v)
    call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    bad_calls, bad_elems, shown, ms = 0, 0, 0, 0.
    t0 = time.time()
    for r in range(reps):
        e0.record()
        got = call()
        e1.record()
        ne = got != ref
        nb = int(ne.sum())
        ms += e0.elapsed_time(e1)
        if nb:
            bad_calls += 1
            bad_elems += nb
            if shown < 3:
                shown += 1
                idx = torch.nonzero(ne.reshape(-1)).reshape(-1)
                L = ref.shape[-1]
                rows = torch.unique(idx // L)
                col = idx % L
                h64 = torch.bincount(col % 64, minlength=64).cpu().tolist()
                h16 = torch.bincount((col % 64) // 16, minlength=4).cpu().tolist()
                d = (got - ref).abs().reshape(-1)[idx]
                print('  launch %d: %d elements in %d points differ; light %% 64 histogram by 16-lane group %s; '
                      'max |diff| %.3g; first (point, light, want, got): %s' % (
                          r, nb, rows.numel(), h16, float(d.max()),
                          [(int(i // L), int(i % L), float(ref.reshape(-1)[i]), float(got.reshape(-1)[i])) for i in idx[:4]]))
    for k in list(ref_opts) + list(test_opts):
        _capi.unset_option(k)
    line = dict(kernel=label, points=n, rows_per_launch=int(ref.numel()), launches=reps, launches_with_differences=bad_calls,
                elements_differing=bad_elems, ms_per_launch=ms / reps, wall_s=round(time.time() - t0, 1))
    print(json.dumps(line), flush=True)
    summary.append(line)


for n in sizes:
    rng, lxyz, _, xyz, cam, normal = scene(n, 41, 16)
    z = rng.normal(size=(n, zd)).astype(np.float32)
    layers, out = net128(30, 90, 1)
    lblob = pack(layers, out, _capi.IN_XYZ_LDIR, 1, cuda)
    largs = (dev(xyz, cuda), dev(lxyz, cuda), lblob)
    soak('resident128_kernel<2,0,8> (lvis, default)', lambda: ops.lvis_fwd(*largs), {'lvis_variant': 4}, {'lvis_variant': 8}, n)
    # round 6: the same kernel storing at final rows (what a render with background rays launches): a 60 % subset of a 5/3 n buffer
    _capi.set_option('lvis_rows', 1)
    n_all = n * 5 // 3
    out_row = torch.from_numpy(np.sort(np.random.default_rng(7).choice(n_all, n, replace=False)).astype(np.int32)).to(cuda)
    full = torch.zeros((n_all, lxyz.shape[0]), device=cuda)
    soak('resident128_kernel<2,0,8,true> (lvis at final rows)', lambda: ops.lvis_fwd(*largs, out=full, out_row=out_row).clone(),
         {'lvis_variant': 4}, {'lvis_variant': 8}, n)
    layers, out = net128(40 + zd, zd + 15, 1)
    bblob = pack(layers, out, _capi.IN_Z_RUSINK, 1, cuda, z_dim=zd)
    bargs = (dev(xyz, cuda), dev(cam, cuda), dev(normal, cuda), dev(z, cuda), dev(lxyz, cuda), bblob)
    soak('brdf_compact_kernel<2,1,8> (opt-in)', lambda: ops.brdf_spec_fwd(*bargs), {'brdf_variant': 6, 'brdf_ct': 4},
         {'brdf_variant': 6, 'brdf_ct': 8}, n)
    if '--geo0' in sys.argv:
        soak('brdf_compact_kernel<2,0,8> (experiment build)', lambda: ops.brdf_spec_fwd(*bargs),
             {'brdf_variant': 5, 'brdf_ct': 4}, {'brdf_variant': 5, 'brdf_ct': 8}, n)
print(json.dumps({'device': torch.cuda.get_device_name(0), 'lib': _capi.LIB_PATH, 'summary': summary}))
