"""Rates of the runtime-shaped MLP kernels (csrc/mlp_generic.hip) next to the tuned kernels on the SAME architectures:
the generality tax of DESIGN.md §3.6.  Prints one JSON object (ms per call, TFLOP/s of the MLP's 2 x MACs)."""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import ops  # noqa: E402

cuda = torch.device('cuda:0')


def timed(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3


def net(d_in, widths, acts, skip_at, rng, prec='bf16'):
    ks, bs, prev = [], [], d_in
    for i, w in enumerate(widths):
        lim = np.sqrt(6. / (prev + w))
        ks.append(rng.uniform(-lim, lim, size=(prev, w)).astype(np.float32))
        bs.append(np.zeros(w, np.float32))
        prev = w + (d_in if skip_at and i in skip_at else 0)
    macs = sum(k.size for k in ks)
    return ops.GenericNet(ks, bs, acts, skip_at, train=True, prec=prec).to(cuda), ks, bs, macs


def main():
    """--only name[,name]: those cases only; --option key=value: a library option (A / B runs, e.g. wgrad_map=0)."""
    rng = np.random.default_rng(0)
    out = {}
    only = sys.argv[sys.argv.index('--only') + 1].split(',') if '--only' in sys.argv else None
    for i, a in enumerate(sys.argv):
        if a == '--option':
            k, v = sys.argv[i + 1].split('=')
            ops._capi.set_option(k, int(v))
    for name, d_in, widths, acts, skip, n, prec in (
            ('nerf_enc_256x8', 63, [256] * 8, ['relu'] * 8, [4], 1 << 18, 'bf16'),
            ('surface_128x4_lvis', 90, [128] * 4 + [1], ['relu'] * 4 + ['sigmoid'], [2], 1 << 21, 'bf16'),
            ('narrow_64x4', 39, [64] * 4 + [4], ['relu'] * 4 + [None], [1], 1 << 20, 'bf16'),
            # 'fp32' = fp32-class: bf16 hi / lo operand pairs (round 5); 'fp32_native' = v_mfma_f32_32x32x2_f32
            ('nerf_enc_256x8_fp32', 63, [256] * 8, ['relu'] * 8, [4], 1 << 17, 'fp32'),
            ('surface_128x4_lvis_fp32', 90, [128] * 4 + [1], ['relu'] * 4 + ['sigmoid'], [2], 1 << 19, 'fp32'),
            ('brdf_prior_128x4_fp32', 18, [128] * 4 + [1], ['relu'] * 4 + ['softplus'], [2], 1 << 19, 'fp32'),
            ('nerf_enc_256x8_fp32_native', 63, [256] * 8, ['relu'] * 8, [4], 1 << 17, 'fp32_native'),
            ('surface_128x4_lvis_fp32_native', 90, [128] * 4 + [1], ['relu'] * 4 + ['sigmoid'], [2], 1 << 19, 'fp32_native')):
        if only and name not in only:
            continue
        g, ks, bs, macs = net(d_in, widths, acts, skip, rng, prec)
        x = torch.randn((n, d_in), device=cuda)
        dy = torch.randn((n, widths[-1]), device=cuda)
        dks = [torch.zeros(k.shape, device=cuda) for k in ks]
        dbs = [torch.zeros(b.shape, device=cuda) for b in bs]
        f = timed(lambda: ops.mlp_generic_fwd(x, g))
        b = timed(lambda: ops.mlp_generic_bwd(x, g, dy, dks, dbs))
        bx = timed(lambda: ops.mlp_generic_bwd(x, g, dy, dks, dbs, want_dx=True))
        out[name] = {'rows': n, 'macs_per_row': macs, 'fwd_ms': round(f, 3), 'fwd_tflops': round(2 * macs * n / f / 1e9, 1),
                     'bwd_ms': round(b, 3), 'bwd_tflops': round(6 * macs * n / b / 1e9, 1),     # fwd recompute + dgrad + wgrad
                     'bwd_with_dx_ms': round(bx, 3),
                     'workspace_mb': round(ops.lib.nfx_mlp_generic_bwd_workspace_bytes(n, g.d_in, g.n_layers, g._w, g._s, g.prec) / 2 ** 20, 1)}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
