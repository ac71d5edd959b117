"""CPU suite, NeRF stage: oracle self-consistency, golden anchors, host packer vs lane-level
kernel emulation, C-ABI symbol export.  No GPU, no compute calls into libnfx."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import nerf_ref
from tests import common, emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_anchors.npz'))


# ------------------------------------------------------------------------------- oracle
def test_embed_layout_and_dims():
    x = np.array([[0.1, -0.2, 0.3]], np.float64)
    e = nerf_ref.embed(x, 10)
    assert e.shape == (1, 63)
    np.testing.assert_allclose(e[0, :3], x[0])
    np.testing.assert_allclose(e[0, 3:6], np.sin(x[0]))
    np.testing.assert_allclose(e[0, 6:9], np.cos(x[0]))
    np.testing.assert_allclose(e[0, 57:60], np.sin(512 * x[0]))
    np.testing.assert_allclose(e[0, 60:63], np.cos(512 * x[0]))
    assert nerf_ref.embed(x, 4).shape == (1, 27) and nerf_ref.embed(x, 2).shape == (1, 15)


def test_mlp_skip_concat_order():
    rng = np.random.default_rng(0)
    x = rng.normal(size=(5, 7))
    layers = [(rng.normal(size=(7, 4)), rng.normal(size=4)),
              (rng.normal(size=(11, 3)), rng.normal(size=3))]
    y = nerf_ref.mlp(x, layers, ['relu', None], skip_at=[0])
    h = np.maximum(x @ layers[0][0] + layers[0][1], 0)
    ref = np.concatenate([h, x], -1) @ layers[1][0] + layers[1][1]  # (y, x): y first (mlp.py:48)
    np.testing.assert_allclose(y, ref)


def test_nerf_param_and_mac_counts():
    net = nerf_ref.init_nerf_net(np.random.default_rng(0))
    n_params = sum(k.size + b.size for layers in net.values() for k, b in layers)
    n_macs = sum(k.size for layers in net.values() for k, _ in layers)
    assert n_params == 595844
    assert n_macs == 593408


def test_gen_z_and_weights_properties():
    rng = np.random.default_rng(1)
    z = nerf_ref.gen_z(2., 6., 64, 5)
    assert z.shape == (5, 64) and z[0, 0] == 2. and abs(z[0, -1] - 6.) < 1e-5
    u = rng.uniform(size=(5, 64)).astype(np.float32)
    zp = nerf_ref.gen_z(2., 6., 64, 5, u=u)
    assert np.all(np.diff(zp, axis=1) >= 0) and zp.min() >= 2. and zp.max() <= 6.
    sigma = rng.normal(size=(5, 64)).astype(np.float32) * 3
    rayd = nerf_ref.l2_normalize(rng.normal(size=(5, 3)).astype(np.float32), 1, 1e-12)
    w = nerf_ref.accumulate_sigma(sigma, z, rayd)
    assert np.all(w >= 0) and np.all(w.sum(-1) <= 1 + 1e-3)
    zl = nerf_ref.gen_z(2., 6., 8, 1, lin_in_disp=True)
    np.testing.assert_allclose(1 / zl[0], np.linspace(1 / 2., 1 / 6., 8), rtol=1e-5)


def test_inverse_transform_sampling_properties():
    rng = np.random.default_rng(2)
    z = nerf_ref.gen_z(2., 6., 64, 7)
    w = rng.uniform(size=(7, 64)).astype(np.float32) ** 4
    z_all = nerf_ref.gen_z_fine(z, w, 128)
    assert z_all.shape == (7, 192) and np.all(np.diff(z_all, axis=1) >= 0)
    mid = .5 * (z[:, 1:] + z[:, :-1])
    zf = nerf_ref.inv_transform_sample(mid, w[:, 1:-1], 128)
    assert np.all(zf >= mid[:, :1] - 1e-6) and np.all(zf <= mid[:, -1:] + 1e-6)
    # all mass in one bin -> every sample inside that bin
    w1 = np.zeros((1, 64), np.float32)
    w1[0, 20] = 1.
    zf = nerf_ref.inv_transform_sample(mid[:1], w1[:, 1:-1], 16)
    assert np.all(zf[:, :-1] >= mid[0, 19] - 1e-6) and np.all(zf[:, :-1] <= mid[0, 20] + 1e-6)
    # u = 1 exceeds cdf[-1] = 1/(1+1e-5): reference behaviour is the LAST midpoint (math.py:84-93)
    assert zf[0, -1] == mid[0, -1]
    # zero weights (empty ray): finite, inside range
    zf = nerf_ref.inv_transform_sample(mid[:1], np.zeros((1, 62), np.float32), 16)
    assert np.all(np.isfinite(zf))
    # fp32 vs fp64 restatement agree
    z64 = nerf_ref.gen_z_fine(z.astype(np.float64), w.astype(np.float64), 128)
    assert np.max(np.abs(z64 - z_all)) < 5e-3


def test_render_rays_fp32_vs_fp64():
    nets = common.nerf_nets(seed=3)
    rayo, rayd = common.camera_rays(4, 4)
    c32, f32, _ = nerf_ref.render_rays(rayo, rayd, nets[0], nets[1])
    nets64 = [{k: [(w.astype(np.float64), b.astype(np.float64)) for w, b in v] for k, v in n.items()}
              for n in nets]
    c64, f64, _ = nerf_ref.render_rays(rayo.astype(np.float64), rayd.astype(np.float64), *nets64)
    for k in ('rgb', 'occu', 'depth'):
        assert np.max(np.abs(c32[k] - c64[k])) < 2e-3, k
        assert np.max(np.abs(f32[k] - f64[k])) < 2e-2, k  # resampling amplifies fp32 cdf noise
    assert 0.05 < float(np.mean(c32['occu'])) < 1.0  # the opaque variant really is non-trivial


def test_psnr_matches_reference_golden():
    v = nerf_ref.psnr_uint8_luma(GOLD['psnr_im1'], GOLD['psnr_im2'])
    assert abs(v - float(GOLD['psnr_value'])) < 1e-9


def test_bf16_round_is_rne():
    x = np.array([1.0, 1.00390625, 1.005859375, 1.001953125, -2.5, 3.14159], np.float32)
    r = nerf_ref.bf16_round(x)
    assert r[0] == 1.0 and r[1] == np.float32(1.0) and r[2] == np.float32(1.0078125)
    assert np.all(np.abs(r - x) <= np.abs(x) * 2 ** -8)


# ------------------------------------------------------------------ C-ABI, packer, emulator
def _header_symbols():
    text = open(os.path.join(ROOT, 'include', 'nfx.h')).read()
    return sorted(set(re.findall(r'\b(nfx_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_header_symbol(nfx_lib):
    syms = _header_symbols()
    assert len(syms) >= 15
    for s in syms:
        assert hasattr(nfx_lib.lib, s), "libnfx.so does not export %s" % s
        assert s in nfx_lib.SIGNATURES, "nerfactor_amd/_capi.py does not bind %s" % s
    assert nfx_lib.lib.nfx_version() >= 100


def test_integration_guide_names_every_header_symbol():
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    missing = [s for s in _header_symbols() if s not in doc]
    assert not missing, "INTEGRATION.md does not mention %s" % missing


def test_pack_rejects_bad_arguments(nfx_lib):
    from nerfactor_amd import ops
    net = common.nerf_nets()[0]
    ks, bs = common.nerf_layers(net)
    with pytest.raises(nfx_lib.NfxError):
        ops.pack_nerf_weights(ks[:-1], bs[:-1])
    with pytest.raises(nfx_lib.NfxError):
        ops.pack_nerf_weights([k.T for k in ks], bs)
    rc = nfx_lib.lib.nfx_nerf_pack_weights(None, None, 0, None, 0)
    assert rc == -1 and 'null' in nfx_lib.last_error()


def test_ops_refuse_cpu_tensors(nfx_lib):
    import torch
    from nerfactor_amd import ops
    with pytest.raises(nfx_lib.NfxError):
        ops.l2_normalize3(torch.zeros(4, 3), 1e-12)


def test_packed_blob_through_lane_emulation_matches_oracle(nfx_lib):
    """Host packer + the kernel's register dataflow (emulated lane by lane) == oracle with the
    same bf16 operand rounding."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(5)
    net = common.nerf_nets(seed=4)[0]
    ks, bs = common.nerf_layers(net)
    blob = ops.pack_nerf_weights(ks, bs, 'bf16').numpy()
    assert blob.nbytes == nfx_lib.lib.nfx_nerf_packed_bytes(0) == 1272 * 1024 + 2496 * 4
    pts = rng.uniform(-3, 3, size=(32, 3)).astype(np.float32)
    views = nerf_ref.l2_normalize(rng.normal(size=(32, 3)).astype(np.float32), 1, 1e-12)
    got = emu.nerf_tile(blob, pts, views)
    want = nerf_ref.eval_nerf_at(pts[:, None, :], views[:, None, :], net,
                                 quant=nerf_ref.bf16_round)[:, 0, :]
    np.testing.assert_allclose(got, want, atol=2e-3, rtol=2e-3)
    # and the bf16 path is a small perturbation of the fp32 reference
    want32 = nerf_ref.eval_nerf_at(pts[:, None, :], views[:, None, :], net)[:, 0, :]
    assert np.max(np.abs(got - want32)) < 0.15


def test_synth_matches_test_generators():
    """bench.py's product-side generators (nerfactor_amd/synth.py, no oracle import) produce the arrays of the
    oracle-based test generators, so bench numbers and test parity refer to the same weights and rays."""
    from nerfactor_amd import synth
    for a, b in zip(synth.nerf_nets(0), common.nerf_nets(0)):
        for name in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
            for (k1, b1), (k2, b2) in zip(a[name], b[name]):
                assert np.array_equal(k1, k2) and np.array_equal(b1, b2), name
    ro, rd = synth.camera_rays(37, 41, cam_loc=(1., 2., 3.))
    ro2, rd2 = common.camera_rays(37, 41, cam_loc=(1., 2., 3.))
    assert np.array_equal(ro, ro2) and np.allclose(rd, rd2, rtol=0, atol=1e-6)
    import ast
    src = open(synth.__file__).read()
    mods = {n.module if isinstance(n, ast.ImportFrom) else a.name for n in ast.walk(ast.parse(src))
            if isinstance(n, (ast.Import, ast.ImportFrom)) for a in n.names}
    assert not any(m and m.split('.')[0] in ('oracle', 'tests') for m in mods), mods


def test_split_operand_arithmetic_of_the_fp32_class_kernels():
    """NFX_PREC_FP32 (mlp_x3.hpp): every operand a bf16 pair hi + lo, every product a_lo b_hi + a_hi b_lo + a_hi b_hi in
    fp32.  Emulated in NumPy: the pair carries 16 significant bits (relative error <= 2^-16), a 256-term dot product
    lands within 1e-5 of float64 relative to sum |a||b| (measured 1.9e-6) — several hundred times closer than single
    bf16 operands (8.8e-4) — and dropping the lo x lo term costs less than 2^-17 of that scale."""
    rng = np.random.default_rng(0)
    x = (rng.normal(size=100000) * np.exp(rng.uniform(-8, 8, size=100000))).astype(np.float32)
    pair = nerf_ref.bf16_pair_round(x)
    assert np.max(np.abs(pair - x) / np.abs(x)) <= 2. ** -16
    assert np.max(np.abs(nerf_ref.bf16_round(x) - x) / np.abs(x)) > 2. ** -9.1       # a single bf16: 8 bits
    a = rng.normal(size=(512, 256)).astype(np.float32)
    b = rng.normal(size=(256, 64)).astype(np.float32)
    a_hi, b_hi = nerf_ref.bf16_round(a), nerf_ref.bf16_round(b)
    a_lo, b_lo = nerf_ref.bf16_round(a - a_hi), nerf_ref.bf16_round(b - b_hi)
    three = (a_lo @ b_hi + a_hi @ b_lo) + a_hi @ b_hi                                  # fp32 accumulation, small terms first
    exact = a.astype(np.float64) @ b.astype(np.float64)
    scale = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    assert np.max(np.abs(three - exact) / scale) < 1e-5
    assert np.max(np.abs(a_hi @ b_hi - exact) / scale) > 100 * np.max(np.abs(three - exact) / scale)   # the bf16 path
    assert np.max(np.abs(a_lo.astype(np.float64) @ b_lo.astype(np.float64)) / scale) < 2. ** -17
