"""CPU suite, NeRFactor stage: oracle vs the importable reference pieces (golden anchors), oracle
self-consistency, and the width-128 packers through the lane-level kernel emulation."""
import os

import numpy as np
import pytest

from oracle import nerf_ref, nerfactor_ref as R
from tests import emu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'reference_anchors.npz'))


# ---------------------------------------------------------------- golden anchors (real reference)
@pytest.mark.parametrize("h", [16, 4])
def test_gen_light_xyz_matches_reference(h):
    xyz, areas = R.gen_light_xyz(h, 2 * h)
    np.testing.assert_allclose(xyz, GOLD['lxyz_%d' % h], rtol=0, atol=1e-10)
    np.testing.assert_allclose(areas, GOLD['lareas_%d' % h], rtol=1e-13)
    assert abs(areas.sum() - 4 * np.pi) < 1e-12


def test_sph2cart_convention_matches_reference():
    sph = GOLD['sph_in']
    r, lat, lng = sph[:, 0], sph[:, 1], sph[:, 2]
    cart = np.stack((r * np.cos(lat) * np.cos(lng), r * np.cos(lat) * np.sin(lng), r * np.sin(lat)), -1)
    np.testing.assert_allclose(cart, GOLD['sph_cart'], atol=1e-12)


def test_dir2rusink_matches_nielsen2015on():
    got = R.dir2rusink(GOLD['rusink_a'], GOLD['rusink_b'])
    np.testing.assert_allclose(got, GOLD['rusink_out'], atol=5e-6)  # eps=1e-6 in safe_l2_normalize
    got32 = R.dir2rusink(GOLD['rusink_a'].astype(np.float32), GOLD['rusink_b'].astype(np.float32))
    d = np.abs(got32 - GOLD['rusink_out'])
    d[:, 0] = np.minimum(d[:, 0], np.pi - d[:, 0])  # phi_d wraps at pi
    assert d.max() < 2e-3


def test_linear2srgb_and_luma_match_xiuminglib():
    np.testing.assert_allclose(R.linear2srgb(GOLD['srgb_in']), GOLD['srgb_out'], atol=1e-12)
    lum = GOLD['psnr_im1'] @ np.array([0.2126, 0.7152, 0.0722])
    np.testing.assert_allclose(lum, GOLD['lum'], atol=1e-12)


# -------------------------------------------------------------------------------- oracle props
def test_world2local_is_orthonormal_and_maps_normal_to_z():
    rng = np.random.default_rng(0)
    n = rng.normal(size=(100, 3))
    rot = R.gen_world2local(n)
    eye = np.einsum('nij,nkj->nik', rot, rot)
    np.testing.assert_allclose(eye, np.broadcast_to(np.eye(3), eye.shape), atol=1e-5)
    nz = np.einsum('nij,nj->ni', rot, nerf_ref.l2_normalize(n, 1, 1e-6))
    np.testing.assert_allclose(nz, np.broadcast_to([0, 0, 1.], nz.shape), atol=1e-5)


def test_microfacet_properties():
    rng = np.random.default_rng(1)
    n_pts, n_l = 50, 64
    normal = nerf_ref.l2_normalize(rng.normal(size=(n_pts, 3)), 1, 1e-6)
    l = nerf_ref.l2_normalize(rng.normal(size=(n_pts, n_l, 3)), 2, 1e-6)
    v = nerf_ref.l2_normalize(rng.normal(size=(n_pts, 3)), 1, 1e-6)
    albedo = rng.uniform(.03, .8, size=(n_pts, 3))
    rough = rng.uniform(.05, 1., size=(n_pts, 1))
    brdf = R.microfacet(l, v, normal, albedo, rough)
    assert brdf.shape == (n_pts, n_l, 3) and np.all(np.isfinite(brdf))
    assert np.all(brdf >= albedo[:, None, :] / np.pi - 1e-12)  # glossy term is non-negative
    lam = R.microfacet(l, v, normal, albedo, rough, lambert_only=True)
    np.testing.assert_allclose(lam, np.broadcast_to((albedo / np.pi)[:, None], lam.shape))
    # half vector below the surface -> D = 0
    hv = nerf_ref.l2_normalize(l + v[:, None, :], 2, 1e-6)
    below = np.einsum('ijk,ik->ij', hv, normal) <= 0
    assert np.allclose((brdf - lam)[below], 0)


def test_render_matches_manual_sum_and_masks_back_lit():
    rng = np.random.default_rng(2)
    lxyz, areas = R.gen_light_xyz(4, 8)
    lxyz = lxyz.reshape(-1, 3)
    pts = rng.uniform(-1, 1, size=(6, 3))
    surf2l = R.calc_ldir(pts, lxyz)
    normal = nerf_ref.l2_normalize(rng.normal(size=(6, 3)), 1, 1e-6)
    brdf = rng.uniform(0, .3, size=(6, 32, 3))
    lvis = rng.uniform(size=(6, 32))
    light = rng.uniform(0, 1, size=(4, 8, 3))
    rgb = R.integrate(brdf, lvis, surf2l, normal, light, areas, to_srgb=False)
    cos = np.einsum('ijk,ik->ij', surf2l, normal)
    manual = np.zeros((6, 3))
    for l in range(32):
        if True:
            manual += (brdf[:, l] * (lvis[:, l] * (cos[:, l] > 0))[:, None] * light.reshape(-1, 3)[l] *
                       cos[:, l, None] * areas.reshape(-1)[l])
    np.testing.assert_allclose(rgb, np.clip(manual, 0, 1), atol=1e-12)


# ------------------------------------------------------------------ packers via lane emulation
def _net128(seed, in_dims, out_dims):
    rng = np.random.default_rng(seed)
    layers, out = R.init_mlp128(rng, in_dims, out_dims)
    for lst in (layers, out):
        for i, (k, b) in enumerate(lst):
            lst[i] = (k, rng.uniform(-.2, .2, size=b.shape).astype(np.float32))
    return layers, out


def _pack(nfx_lib, layers, out, in_kind, out_dim, z_dim=0):
    from nerfactor_amd import ops
    ks = [k for k, _ in layers] + [out[0][0]]
    bs = [b for _, b in layers] + [out[0][1]]
    return ops.pack_mlp128_weights(ks, bs, in_kind, out_dim, z_dim=z_dim).numpy()


def test_mlp128_xyz_pack_through_emulation(nfx_lib):
    layers, out = _net128(3, 63, 3)
    blob = _pack(nfx_lib, layers, out, nfx_lib.IN_XYZ, 3)
    assert blob.nbytes == 136 * 1024 + 544 * 4
    pts = np.random.default_rng(4).uniform(-1.5, 1.5, size=(32, 3)).astype(np.float32)
    got = emu.mlp128_xyz_tile(blob, pts, 3)
    want = R.mlp128(nerf_ref.embed(pts, 10), layers, out, None, quant=nerf_ref.bf16_round)
    np.testing.assert_allclose(got, want, atol=2e-3, rtol=2e-3)


def test_lvis_pack_through_emulation(nfx_lib):
    layers, out = _net128(5, 90, 1)
    blob = _pack(nfx_lib, layers, out, nfx_lib.IN_XYZ_LDIR, 1)
    assert blob.nbytes == (32 * 1024 + 1024) + (136 * 1024 + 544 * 4)
    rng = np.random.default_rng(6)
    pt = rng.uniform(-1, 1, size=3).astype(np.float32)
    ldirs = nerf_ref.l2_normalize(rng.normal(size=(32, 3)).astype(np.float32), 1, 1e-6)
    got = emu.lvis_tile(blob, pt, ldirs)
    x = np.concatenate((nerf_ref.embed(np.broadcast_to(pt, (32, 3)), 10), nerf_ref.embed(ldirs, 4)), -1)
    want = R.mlp128(x, layers, out, None, quant=nerf_ref.bf16_round)[:, 0]
    # the per-point fold keeps the posenc(xyz) partial sums in fp32 (more accurate than rounding
    # the concatenated input): compare against both the rounded and the fp32 evaluation
    want32 = R.mlp128(x, layers, out, None)[:, 0]
    assert np.max(np.abs(got - want)) < 1.5e-2 and np.max(np.abs(got - want32)) < 3e-2


@pytest.mark.parametrize("zd", [3, 1, 6])
def test_brdf_pack_through_emulation(nfx_lib, zd):
    layers, out = _net128(7, zd + 15, 1)
    blob = _pack(nfx_lib, layers, out, nfx_lib.IN_Z_RUSINK, 1, z_dim=zd)
    rng = np.random.default_rng(8)
    z = rng.normal(size=(32, zd)).astype(np.float32)
    rus = rng.uniform(0, np.pi / 2, size=(32, 3)).astype(np.float32)
    got = emu.brdf_tile(blob, z, rus)
    x = np.concatenate((z, nerf_ref.embed(rus, 2)), -1)
    want = R.mlp128(x, layers, out, None, quant=nerf_ref.bf16_round)[:, 0]
    np.testing.assert_allclose(got, want, atol=2e-3, rtol=2e-3)


@pytest.mark.parametrize("variant", ['microfacet', 'learned'])
def test_torch_port_of_the_render_matches_the_numpy_oracle(variant):
    """oracle/torch_ref.py:nerfactor_render (the timed CPU baseline and PSNR reference of bench.py's NeRFactor legs)
    against oracle/nerfactor_ref.py:nerfactor_call, which is the restatement pinned to the reference fixtures."""
    import torch
    from oracle import torch_ref as T
    rng = np.random.default_rng(0)
    n = 48
    net = R.init_nerfactor_net(rng, 1 if variant == 'microfacet' else 3)
    for k in net:
        net[k] = [(w, rng.uniform(-.2, .2, size=b.shape).astype(np.float32)) for w, b in net[k]]
    bn = R.init_brdf_mlp(rng, 3)
    lxyz, lareas = R.gen_light_xyz(16, 32)
    lxyz, lareas = lxyz.astype(np.float32), lareas.astype(np.float32)
    xyz = rng.uniform(-1, 1, (n, 3)).astype(np.float32)
    alpha = (rng.uniform(size=(n, 1)) < .6).astype(np.float32)
    rayo = np.broadcast_to(np.array([2.2, -2.4, 1.7], np.float32), (n, 3)).copy()
    light = rng.uniform(size=(16, 32, 3)).astype(np.float32)
    probes = [np.exp(rng.normal(size=(16, 32, 3))).astype(np.float32) for _ in range(2)]
    batch = (rayo, np.zeros((n, 3), np.float32), alpha, xyz, np.ones((n, 3), np.float32), np.ones((n, 512), np.float32))
    pred = R.nerfactor_call(batch, net, lxyz, lareas, light, variant=variant, brdf_net=bn, probes=probes)[0]
    t = lambda d: {k: [(torch.from_numpy(w), torch.from_numpy(b)) for w, b in v] for k, v in d.items()}
    lights = torch.from_numpy(np.stack([light] + probes).reshape(3, -1, 3))
    out = T.nerfactor_render((torch.from_numpy(rayo), torch.from_numpy(alpha), torch.from_numpy(xyz)), t(net),
                             torch.from_numpy(lxyz), torch.from_numpy(lareas), lights, variant=variant, brdf_net=t(bn))
    assert np.abs(out['rgb'][:, 0].numpy() - pred['rgb']).max() < 2e-5
    assert np.abs(out['rgb'][:, 1:].numpy() - pred['rgb_probes']).max() < 2e-5
    for k in ('normal', 'lvis', 'albedo', 'brdf'):
        assert np.abs(out[k].numpy() - pred[k]).max() < 5e-6, k
    assert (out['rgb'][alpha[:, 0] == 0] == 0).all()


def test_hdr_probe_reader_on_hand_assembled_rgbe_bytes(tmp_path):
    """util/light.py:read_hdr against bytes assembled here from the Radiance RGBE specification (not by the module's own
    writer): new-style run-length scanlines (marker 2 2 hi lo, per-channel runs > 128 / literals <= 128), a flat
    scanline, exponent 0 = black, OpenCV's decoding m * 2^(e - 136) that xm.io.hdr.read returns (io/hdr.py:11-27)."""
    from nerfactor_amd.nerfactor.util import light as L
    w = 10
    # scanline 0 (RLE): R = run of 10 x 128; G = literal 0..9; B = run of 4 x 7 then literal 6 values; E = run 10 x 129
    rle = bytes([2, 2, 0, w]) + bytes([128 + 10, 128]) + bytes([10] + list(range(10))) + \
        bytes([128 + 4, 7, 6, 1, 2, 3, 4, 5, 6]) + bytes([128 + 10, 129])
    # scanline 1 (RLE): everything literal, exponents alternate 0 (black) and 136
    r1 = list(range(100, 110))
    rle += bytes([2, 2, 0, w]) + bytes([10] + r1) + bytes([10] + r1) + bytes([10] + r1) + \
        bytes([10] + [0, 136] * 5)
    path = str(tmp_path / 'p.hdr')
    with open(path, 'wb') as h:
        h.write(b'#?RADIANCE\n# made by hand\nFORMAT=32-bit_rle_rgbe\nEXPOSURE=1.0\n\n-Y 2 +X 10\n' + rle)
    img = L.read_hdr(path)
    assert img.shape == (2, w, 3) and img.dtype == np.float32
    f = 2. ** (129 - 136)
    np.testing.assert_array_equal(img[0, :, 0], np.full(w, 128 * f, np.float32))
    np.testing.assert_array_equal(img[0, :, 1], np.arange(10, dtype=np.float32) * f)
    np.testing.assert_array_equal(img[0, :, 2], np.float32([7, 7, 7, 7, 1, 2, 3, 4, 5, 6]) * f)
    np.testing.assert_array_equal(img[1, 0::2], 0.)                                  # exponent byte 0
    np.testing.assert_array_equal(img[1, 1::2, 0], np.float32(r1[1::2]))             # 2^(136 - 136) = 1
    # a width below 8 is always stored flat
    with open(path, 'wb') as h:
        h.write(b'#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y 1 +X 2\n' + bytes([1, 2, 3, 137, 4, 5, 6, 135]))
    np.testing.assert_array_equal(L.read_hdr(path), np.float32([[[2, 4, 6], [2, 2.5, 3]]]))


def test_probe_resize_is_the_antialiased_triangle_filter():
    """tf.image.resize(bilinear, antialias=True) (util/img.py:129-130) = a triangle kernel widened by the down-scaling
    factor on half-pixel centres; torch's antialiased bilinear interpolation implements the same filter and serves as
    the independent check.  Constant images stay constant, the mean is preserved."""
    import torch
    from nerfactor_amd.nerfactor.util import light as L
    rng = np.random.default_rng(3)
    for h in (64, 50, 16, 8):
        img = np.exp(rng.normal(size=(h, 2 * h, 3))).astype(np.float32)
        got = L.resize_antialias(img, new_h=16)
        want = torch.nn.functional.interpolate(torch.from_numpy(img).permute(2, 0, 1)[None], size=(16, 32),
                                               mode='bilinear', antialias=True, align_corners=False)
        np.testing.assert_allclose(got, want[0].permute(1, 2, 0).numpy(), rtol=2e-6, atol=2e-6)
    assert np.allclose(L.resize_antialias(np.full((40, 80, 3), 2.5, np.float32), new_h=16), 2.5)


def test_model_loads_hdr_probes_from_test_envmap_dir(tmp_path):
    """ADVICE r01 / VERDICT missing #3: a reference config whose test_envmap_dir holds .hdr probes must yield those
    probes (name = file name without extension), resized to light_res."""
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    from nerfactor_amd.nerfactor.util import light as L
    rng = np.random.default_rng(5)
    maps = {'city': np.exp(rng.normal(size=(32, 64, 3))).astype(np.float32),
            'studio': np.exp(rng.normal(size=(16, 32, 3))).astype(np.float32)}
    for name, m in maps.items():
        L.write_hdr(m, str(tmp_path / (name + '.hdr')))
    cfg = make_config('nerfactor_microfacet', shape_mode='scratch', test_envmap_dir=str(tmp_path))
    model = get_model_class('nerfactor_microfacet')(cfg)
    assert list(model.novel_probes) == ['city', 'studio']
    for name, m in maps.items():
        p = model.novel_probes[name].numpy()
        assert p.shape == (16, 32, 3)
        want = L.resize_antialias(L.read_hdr(str(tmp_path / (name + '.hdr'))), new_h=16)
        np.testing.assert_array_equal(p, want)
        assert abs(p.mean() / m.mean() - 1) < 0.02      # RGBE keeps 8 bits of mantissa


def test_check_numerics_is_immediate_outside_training_and_deferred_inside():
    """models/base.py:check_numerics mirrors tf.debugging.check_numerics; inside a training step (autograd recording)
    the verdict is raised by flush_numerics() instead of stalling the launch queue mid-step."""
    import torch
    from nerfactor_amd.nerfactor.models.base import Model

    class M(Model):
        def __init__(self):
            torch.nn.Module.__init__(self)
    m = M()
    bad = torch.tensor([1., float('nan')])
    with torch.no_grad():
        assert m.check_numerics(torch.ones(3), "fine") is not None
        with pytest.raises(FloatingPointError, match="Albedo"):
            m.check_numerics(bad, "Albedo")
    assert m.check_numerics(torch.ones(3), "fine").shape == (3,)
    m.flush_numerics()                       # nothing wrong so far
    m.check_numerics(torch.ones(3), "fine")
    m.check_numerics(bad, "Loss")            # recorded, not raised
    with pytest.raises(FloatingPointError, match="Loss"):
        m.flush_numerics()
    m.flush_numerics()                       # the queue was cleared
