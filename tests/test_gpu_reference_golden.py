"""The HIP path (through the reference-shaped plugin classes) against outputs of the REFERENCE'S OWN model classes:
tests/golden/reference_models.npz, produced by tests/golden/make_reference_golden.py from google/nerfactor's
unmodified Python (run on the NumPy TensorFlow shim in tests/golden/tf_shim).

The reference computes in float32; the kernels multiply in bf16 on the MFMA units (fp32 accumulate), so the stated
tolerance is the bf16 one of SURVEY.md §8d: max-abs <= 3e-2 on [0,1]-valued outputs, PSNR >= 40 dB on the render;
the fp32-class NeRF path (precision = fp32) is held to 2e-4-class bounds."""
import os

import numpy as np
import pytest
import torch

from oracle import nerf_ref
from tests import common
from tests.golden import golden_inputs as gi

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'reference_models.npz'))


def dev(a, cuda):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(cuda)


def fill(network, pairs):
    assert len(network.layers) == len(pairs)
    for layer, (k, b) in zip(network.layers, pairs):
        layer.kernel.data.copy_(torch.from_numpy(k))
        layer.bias.data.copy_(torch.from_numpy(b))


def make(name, cuda, **over):
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    torch.manual_seed(0)
    return get_model_class(name)(make_config(name, **over)).to(cuda)


# ---------------------------------------------------------------------------------------------- NeRF
@pytest.mark.parametrize('prec', ['bf16', 'fp32'])
def test_nerf_plugin_render_vs_reference_outputs(nfx_lib, cuda, prec):
    model = make('nerf', cuda, precision=prec)
    nets = common.nerf_nets(seed=gi.NERF_SEED)
    np.testing.assert_allclose(gi.checksum_nerf(nets), GOLD['nerf_weight_checksum'], rtol=1e-6)
    for pref, net in zip(('coarse_', 'fine_'), nets):
        for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
            fill(model.net[pref + part], net[part])
    rayo, rayd, gt = gi.nerf_rays()
    n = rayo.shape[0]
    batch = (['x'] * n, torch.tensor([[8, 8]] * n), dev(rayo, cuda), dev(rayd, cuda), dev(gt, cuda))
    pred, gt_t, loss_kwargs, to_vis = model(batch, mode='test')
    # rays whose last-sample logit sits inside the rounding noise flip between "hit" and "background"
    # (dist_last = 1e10, nerf.py:186-191).  precision = bf16 (r04): the plugin evaluates that one sample fp32-class
    # (models/nerf.py last_sample_precision), so ALL 64 rays are held to the bound — 8 of them sit inside the 0.06 band
    # rounds 1-3 excused.  precision = fp32: the 1e-2 band of the fp32-class kernel itself (no ray of this fixture inside).
    _, _, aux = nerf_ref.render_rays(rayo, rayd, nets[0], nets[1])
    band = 0. if prec == 'bf16' else 1e-2
    ok_c = np.abs(aux['rgbs_coarse'][:, -1, 3]) >= band
    ok_f = ok_c & (np.abs(aux['rgbs_fine'][:, -1, 3]) >= band)
    assert int(ok_f.sum()) == 64, int(ok_f.sum())
    in_r03_band = (np.abs(aux['rgbs_coarse'][:, -1, 3]) <= 0.06) | (np.abs(aux['rgbs_fine'][:, -1, 3]) <= 0.06)
    assert int(in_r03_band.sum()) == 8, int(in_r03_band.sum())     # an exact count, so a drift of the oracle shows
    tol_rgb, tol_occu, tol_med = (3e-2, 8e-2, 5e-3) if prec == 'bf16' else (2e-3, 2e-3, 1e-4)
    # what bf16 operands alone do to this input: the ORACLE with bf16-rounded operands (fp32 everywhere else)
    cq, fq, _ = nerf_ref.render_rays(rayo, rayd, nets[0], nets[1], quant=nerf_ref.bf16_round)
    for lvl, ok, oq in (('coarse', ok_c, cq), ('fine', ok_f, fq)):
        rgb = to_vis[lvl + '_rgb'].cpu().numpy()
        want = GOLD['nerf_%s_rgb' % lvl]
        err = np.abs(rgb - want).max(-1)
        # The stated tolerance on every ray; a ray above it is excused only — counted, at most 2 % = one ray of the 64 —
        # when the ORACLE with bf16 operands is itself beyond the tolerance on that very ray (measured: one ray, 3.03e-2:
        # the bound is a statement about bf16 operands, which no kernel can beat).  r01-r04 allowed 4/3 of the tolerance
        # on all rays instead.
        oracle_q = np.abs(oq['rgb'] - want).max(-1)
        over = np.flatnonzero(ok & (err > tol_rgb))
        print(prec, lvl, "rays above %.0e: %s (kernel %s, bf16-operand oracle %s)" % (
            tol_rgb, over.tolist(), err[over].round(4).tolist(), oracle_q[over].round(4).tolist()))
        assert len(over) <= 0.02 * ok.sum() and np.all(oracle_q[over] > tol_rgb * 0.95) and np.all(err[over] <= 1.2 * oracle_q[over]), (
            lvl, over.tolist(), err[over].tolist(), oracle_q[over].tolist())
        assert np.median(err) <= tol_med, (lvl, np.median(err))
        occu = to_vis[lvl + '_occu'].cpu().numpy()
        assert np.abs(occu - GOLD['nerf_%s_occu' % lvl])[ok].max() <= tol_occu
        psnr = nerf_ref.psnr_uint8_luma(rgb.reshape(8, 8, 3), want.reshape(8, 8, 3))
        assert psnr >= (40. if prec == 'bf16' else 55.), (lvl, psnr)
    loss = float(model.compute_loss(pred, gt_t, **loss_kwargs))
    # the scalar loss includes the rays on the discontinuity: 1 % in bf16 (rgb errors up to 3e-2 on a 0.3 loss)
    assert abs(loss - float(GOLD['nerf_loss'])) <= (1e-2 if prec == 'bf16' else 2e-4) * float(GOLD['nerf_loss'])


def test_sampler_and_compositing_kernels_vs_reference_outputs(nfx_lib, cuda):
    from nerfactor_amd import ops
    z, w, sigma, rd = gi.sampler_inputs()
    got = ops.sample_fine(dev(z, cuda), dev(w, cuda), 128).cpu().numpy()
    bad = np.abs(got - GOLD['nerf_z_fine']) > 1e-5
    assert bad.mean() < 2e-3 and np.all(np.diff(got, axis=1) >= 0)
    np.testing.assert_allclose(ops.gen_z(2., 6., 64, 3, device=cuda).cpu().numpy(), GOLD['nerf_gen_z'], atol=1e-6)
    raw = np.zeros((32, 64, 4), np.float32)
    raw[..., 3] = sigma
    weights = ops.composite_fwd(dev(raw, cuda), dev(z, cuda), dev(rd, cuda), white_bg=True)[4].cpu().numpy()
    np.testing.assert_allclose(weights, GOLD['nerf_acc_weights'], rtol=1e-4, atol=2e-6)


# ---------------------------------------------------------------------------------------------- shape / BRDF prior
def surface_batch(cuda):
    rayo, rgb, alpha, xyz, normal, lvis = gi.surface_batch(512)
    n = rayo.shape[0]
    return (['x'] * n, torch.tensor([[4, n // 4]] * n), dev(rayo, cuda), dev(np.zeros_like(rayo), cuda),
            dev(rgb, cuda), dev(alpha, cuda), dev(xyz, cuda), dev(normal, cuda), dev(lvis, cuda))


def test_shape_plugin_vs_reference_outputs(nfx_lib, cuda):
    model = make('shape', cuda, xyz_jitter_std='0')
    net = gi.nerfactor_net(3)
    for part in ('normal_mlp', 'normal_out', 'lvis_mlp', 'lvis_out'):
        fill(model.net[part], net[part])
    pred, gt, kw, _ = model(surface_batch(cuda), mode='test')
    assert np.abs(pred['normal'].cpu().numpy() - GOLD['shape_normal']).max() < 3e-2
    assert np.abs(pred['lvis'].cpu().numpy() - GOLD['shape_lvis']).max() < 3e-2
    loss = model.compute_loss(pred, gt, **kw).cpu().numpy()
    np.testing.assert_allclose(loss, GOLD['shape_loss'], rtol=5e-2, atol=2e-3)


def test_brdf_prior_plugin_vs_reference_outputs(nfx_lib, cuda, tmp_path):
    for name in gi.BRDF_NAMES:
        (tmp_path / ('train_%s.npz' % name)).write_bytes(b'')
    model = make('brdf', cuda, data_root=str(tmp_path))
    assert model.brdf_names == gi.BRDF_NAMES
    bnet = gi.brdf_net()
    fill(model.net['brdf_mlp'], bnet['brdf_mlp'])
    fill(model.net['brdf_out'], bnet['brdf_out'])
    model.latent_code.z = gi.latent_codes()
    model.to(cuda)
    i, rusink, refl = gi.brdf_batch()
    n = rusink.shape[0]
    batch = (['x'] * n, torch.from_numpy(i).to(cuda), None, None, None, dev(rusink, cuda), dev(refl, cuda))
    pred, gt, kw, to_vis = model(batch, mode='vali')
    # fused bf16 template (nfx_brdf_rows_fwd): bf16 operand rounding of an 18 -> 128 x 4 -> 1 softplus MLP
    np.testing.assert_allclose(pred['brdf'].cpu().numpy(), GOLD['brdf_pred'], rtol=2e-2, atol=2e-3)
    np.testing.assert_allclose(pred['brdf_reci'].cpu().numpy(), GOLD['brdf_pred_reci'], rtol=2e-2, atol=2e-3)
    np.testing.assert_array_equal(to_vis['z'].cpu().numpy(), GOLD['brdf_z'])
    np.testing.assert_allclose(float(model.compute_loss(pred, gt, **kw)), float(GOLD['brdf_loss']), rtol=2e-2)
    np.testing.assert_allclose(model.latent_code.interp(0.25, 0, 0.75, 2).detach().cpu().numpy(),
                               GOLD['brdf_interp'], rtol=1e-6)


# ---------------------------------------------------------------------------------------------- NeRFactor
@pytest.mark.parametrize('prec', ['bf16', 'fp32'])
@pytest.mark.parametrize('tag', ['nfl', 'nfm'])
def test_nerfactor_plugin_vs_reference_outputs(nfx_lib, cuda, tag, prec):
    """Model.call(mode='test', relight_olat=True) against the REFERENCE's own outputs (reference_models.npz).
    bf16 operands: 3e-2 on every [0, 1]-valued output; `precision = fp32` (fp32-class kernels): 3e-4 on the MLP heads,
    1e-3 on rgb, 6e-3 on the x200 OLAT renders — held to the reference fixtures themselves, not to the oracle's float64
    re-evaluation (the CPU oracle sits within 1e-5 / 1e-4 of these fixtures, tests/test_cpu_reference_golden.py)."""
    learned = tag == 'nfl'
    name = 'nerfactor' if learned else 'nerfactor_microfacet'
    tol_head, tol_rgb, tol_olat = (3e-2, 3e-2, 6e-2) if prec == 'bf16' else (3e-4, 1e-3, 6e-3)
    model = make(name, cuda, shape_mode='finetune', shape_model_ckpt='none', brdf_model_ckpt='none',
                 test_envmap_dir='', precision=prec)
    net = gi.nerfactor_net(3 if learned else 1)
    for part in net:
        fill(model.net[part], net[part])
    if learned:
        bnet = gi.brdf_net()
        fill(model.brdf_model.net['brdf_mlp'], bnet['brdf_mlp'])
        fill(model.brdf_model.net['brdf_out'], bnet['brdf_out'])
    model._light.data.copy_(torch.from_numpy(gi.light_probe(gi.LIGHT_SCALE[tag])))
    pred, gt, kw, _ = model(surface_batch(cuda), mode='test', relight_olat=True)
    errs = {}
    for k in ('normal', 'lvis', 'albedo', 'brdf', 'rgb'):
        errs[k] = float(np.abs(pred[k].cpu().numpy() - GOLD['%s_test_%s' % (tag, k)]).max())
        assert errs[k] < (tol_rgb if k == 'rgb' else tol_head), (k, errs[k])
    keys = [str(k) for k in GOLD['%s_olat_keys' % tag]]
    idx = [int(k[:4]) * 32 + int(k[5:]) for k in keys]
    olat = pred['rgb_olat'].cpu().numpy()
    assert olat.shape == (24, 512, 3)
    errs['olat'] = float(np.abs(olat[:, idx] - GOLD['%s_test_rgb_olat' % tag]).max())
    assert errs['olat'] < tol_olat, errs     # one light x 200: steep tonemap
    loss = model.compute_loss(pred, gt, **dict(kw, mode='vali')).cpu().numpy()
    np.testing.assert_allclose(loss, GOLD['%s_vali_loss' % tag], atol=5e-3 if prec == 'bf16' else 2e-4)
    print(tag, prec, 'max-abs vs the reference fixtures:', errs)


@pytest.mark.parametrize('tag', ['nfl', 'nfm'])
def test_nerfactor_plugin_editing_hooks_vs_reference_outputs(nfx_lib, cuda, tag):
    learned = tag == 'nfl'
    name = 'nerfactor' if learned else 'nerfactor_microfacet'
    model = make(name, cuda, shape_mode='finetune', shape_model_ckpt='none', brdf_model_ckpt='none',
                 test_envmap_dir='')
    net = gi.nerfactor_net(3 if learned else 1)
    for part in net:
        fill(model.net[part], net[part])
    if learned:
        bnet = gi.brdf_net()
        fill(model.brdf_model.net['brdf_mlp'], bnet['brdf_mlp'])
        fill(model.brdf_model.net['brdf_out'], bnet['brdf_out'])
    model._light.data.copy_(torch.from_numpy(gi.light_probe(gi.LIGHT_SCALE[tag])))
    scales, ao_flat, ao_map, z_over = gi.edit_inputs(3 if learned else 1)
    for ename, edit in (('scaled', dict(albedo_scales=dev(scales, cuda), brdf_z_override=dev(z_over, cuda))),
                        ('flat_albedo', dict(albedo_override=dev(ao_flat, cuda))),
                        ('albedo_map', dict(albedo_override=dev(ao_map, cuda)))):
        pred = model(surface_batch(cuda), mode='test', **edit)[0]
        for k in ('albedo', 'brdf', 'rgb'):
            err = np.abs(pred[k].cpu().numpy() - GOLD['%s_edit_%s_%s' % (tag, ename, k)]).max()
            assert err < 3e-2, (ename, k, err)


def test_nerfactor_plugin_shape_mode_nerf_vs_reference_outputs(nfx_lib, cuda):
    model = make('nerfactor_microfacet', cuda, shape_mode='nerf', test_envmap_dir='')
    net = gi.nerfactor_net(1)
    for part in ('albedo_mlp', 'albedo_out', 'brdf_z_mlp', 'brdf_z_out'):
        fill(model.net[part], net[part])
    model._light.data.copy_(torch.from_numpy(gi.light_probe(gi.LIGHT_SCALE['nfm'])))
    pred = model(surface_batch(cuda), mode='test')[0]
    for k in ('normal', 'lvis'):
        np.testing.assert_allclose(pred[k].cpu().numpy(), GOLD['nfm_shapenerf_' + k], atol=1e-5)
    assert np.abs(pred['rgb'].cpu().numpy() - GOLD['nfm_shapenerf_rgb']).max() < 3e-2


def _excluded(err, tol, what, max_frac=0.02):
    """Counted exclusion list: every element must be within `tol` except an explicit, reported set of at most
    `max_frac` of them (max-abs bounds on >= 98 % of the elements, no quantile bounds)."""
    bad = np.flatnonzero(err > tol)
    print("%s: %d of %d elements above %.0e (max %.3e): %s" % (what, len(bad), err.size, tol, err.max(), bad[:16].tolist()))
    assert len(bad) <= max_frac * err.size, (what, len(bad), err.size, float(err.max()))
    return bad


# ---------------------------------------------------------------------------------------------- geometry_from_nerf
@pytest.mark.parametrize('bbox', [False, True])
def test_geometry_extraction_vs_reference_outputs(nfx_lib, cuda, bbox):
    """compute_depth_and_normal / compute_light_visibility of the plugin (density-gradient kernel, shadow-ray
    marching) against what the reference's own functions produced for the same NeRF (bf16 bounds of
    tests/test_gpu_nerf.py::test_geometry_extraction_vs_oracle)."""
    from nerfactor_amd.nerfactor import geometry_from_nerf as G
    from nerfactor_amd.nerfactor.config import make_config
    cfg = make_config('nerf')
    model = make('nerf', cuda)
    nets = common.nerf_nets(seed=gi.NERF_SEED)
    for pref, net in zip(('coarse_', 'fine_'), nets):
        for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
            fill(model.net[pref + part], net[part])
    box = tuple(float(v) for v in gi.GEOM_BBOX.split(',')) if bbox else None
    tag = 'geo_bbox_' if bbox else 'geo_'
    rayo, rayd, _ = gi.nerf_rays()
    rayo, rayd = rayo[:gi.GEOM_RAYS], nerf_ref.l2_normalize(rayd[:gi.GEOM_RAYS], 1, 1e-12)
    with torch.no_grad():
        occu, depth, normal = (t.cpu().numpy() for t in
                               G.compute_depth_and_normal(model, dev(rayo, cuda), dev(rayd, cuda), cfg, bbox=box))
    # (r05: max-abs on every ray — the last sample of a ray, whose sign decides the ray, is evaluated fp32-class in the
    #  geometry march too (models/nerf.py:_refine_last_sigma); rounds 1-4 bounded the 0.9-quantile here)
    _excluded(np.abs(occu - GOLD[tag + 'occu']), 3e-2, tag + 'occupancy')
    _excluded(np.abs(depth - GOLD[tag + 'depth']), 0.16, tag + 'depth')            # 4 % of the [2, 6] depth range
    dn = np.abs(normal - GOLD[tag + 'normal']).max(1)
    _excluded(dn, 8e-2, tag + 'normal (as a vector)')
    assert np.median(dn) <= 4e-2, np.median(dn)
    if bbox:
        surf, nrm = gi.geom_bbox_points()
    else:
        surf, nrm = GOLD['geo_surf'], GOLD['geo_surf_normal']
    with torch.no_grad():
        lvis = G.compute_light_visibility(model, dev(surf, cuda), dev(nrm, cuda), cfg, lvis_far=1.,
                                          light_h=gi.GEOM_LIGHT_H, bbox=box).cpu().numpy()
    want = GOLD[tag + 'lvis']
    assert lvis.shape == want.shape
    assert np.mean((lvis == 0) != (want == 0)) < 0.02                       # same front-lit set
    _excluded(np.abs(lvis - want).reshape(-1), 4e-2, tag + 'light visibility')
    assert np.abs(lvis - want).mean() <= 2e-2


# ---------------------------------------------------------------------------------------------- larger fixtures (r02)
@pytest.mark.parametrize('coarse_precision', ['bf16', 'fp32'])
def test_trained_nerf_1024_rays_vs_reference_outputs(nfx_lib, cuda, coarse_precision):
    """1024 rays of a 32 x 32 view through the trained networks against the reference's own render: max-abs 3e-2 on
    rgb / occupancy (4 % of the depth range on depth) for at least 98 % of the rays — silhouette rays, where a bf16-sized
    change of the density moves the accumulated opacity, are the counted exceptions — and PSNR >= 40 dB.
    coarse_precision = fp32 (ini key, r05): the coarse pass fp32-class — the rays above the tolerance are a coarse-pass
    effect (bf16 coarse weights move a silhouette ray's fine samples across the fitted density edge):
    with it EVERY ray's rgb is within 3e-2, no exclusion."""
    model = make('nerf', cuda, coarse_precision=coarse_precision)
    nets = gi.trained_nerf_nets()
    np.testing.assert_allclose(gi.checksum_nerf(nets), GOLD['nerf1k_weight_checksum'], rtol=1e-6)
    for pref, net in zip(('coarse_', 'fine_'), nets):
        for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
            fill(model.net[pref + part], net[part])
    rayo, rayd, gt = gi.nerf1k_rays()
    n, (h, w) = rayo.shape[0], gi.NERF1K_HW
    batch = (['x'] * n, torch.tensor([[h, w]] * n), dev(rayo, cuda), dev(rayd, cuda), dev(gt, cuda))
    pred, gt_t, loss_kwargs, to_vis = model(batch, mode='test')
    for lvl in ('coarse', 'fine'):
        rgb = to_vis[lvl + '_rgb'].cpu().numpy()
        want = GOLD['nerf1k_%s_rgb' % lvl]
        _excluded(np.abs(rgb - want).max(-1), 3e-2, lvl + ' rgb', max_frac=0. if coarse_precision == 'fp32' else 0.02)
        _excluded(np.abs(to_vis[lvl + '_occu'].cpu().numpy() - GOLD['nerf1k_%s_occu' % lvl]), 3e-2, lvl + ' occu')
        _excluded(np.abs(to_vis[lvl + '_depth'].cpu().numpy() - GOLD['nerf1k_%s_depth' % lvl]), 0.16, lvl + ' depth')
        assert np.median(np.abs(rgb - want).max(-1)) < 3e-3
        psnr = nerf_ref.psnr_uint8_luma(rgb.reshape(h, w, 3), want.reshape(h, w, 3))
        assert psnr >= 40., (lvl, psnr)
    loss = float(model.compute_loss(pred, gt_t, **loss_kwargs))
    assert abs(loss - float(GOLD['nerf1k_loss'])) <= 5e-3 * float(GOLD['nerf1k_loss'])


@pytest.mark.parametrize('tag', ['nfl256', 'nfm256'])
def test_nerfactor_256_points_vs_reference_outputs(nfx_lib, cuda, tag):
    learned = tag.startswith('nfl')
    name = 'nerfactor' if learned else 'nerfactor_microfacet'
    model = make(name, cuda, shape_mode='finetune', shape_model_ckpt='none', brdf_model_ckpt='none', test_envmap_dir='')
    net = gi.nerfactor_net(3 if learned else 1)
    for part in net:
        fill(model.net[part], net[part])
    if learned:
        bnet = gi.brdf_net()
        fill(model.brdf_model.net['brdf_mlp'], bnet['brdf_mlp'])
        fill(model.brdf_model.net['brdf_out'], bnet['brdf_out'])
    model._light.data.copy_(torch.from_numpy(gi.light_probe(gi.LIGHT_SCALE['nfl' if learned else 'nfm'])))
    rayo, rgb, alpha, xyz, normal, lvis = gi.surface_batch(512, n=gi.SURF256, seed=37)
    n = rayo.shape[0]
    batch = (['x'] * n, torch.tensor([[16, n // 16]] * n), dev(rayo, cuda), dev(np.zeros_like(rayo), cuda),
             dev(rgb, cuda), dev(alpha, cuda), dev(xyz, cuda), dev(normal, cuda), dev(lvis, cuda))
    pred = model(batch, mode='test')[0]
    for k in ('normal', 'albedo', 'brdf'):
        err = np.abs(pred[k].cpu().numpy() - GOLD['%s_%s' % (tag, k)])
        assert err.max() < 3e-2, (k, err.max())
    err = np.abs(pred['lvis'].cpu().numpy()[:, ::gi.LVIS_STRIDE] - GOLD[tag + '_lvis'])
    assert err.max() < 3e-2, err.max()
    # rgb: points seen at grazing angles are the counted exceptions (the reference divides by 4 |l.n| |v.n|)
    _excluded(np.abs(pred['rgb'].cpu().numpy() - GOLD[tag + '_rgb']).max(1), 3e-2, tag + ' rgb')


def test_geometry_on_the_trained_nerf_vs_reference_outputs(nfx_lib, cuda):
    """geometry_from_nerf on a real surface (the unit sphere the networks were fitted to): max-abs bounds on at least
    98 % of the rays for occupancy / depth / normal and of the (point, light) pairs for the visibility."""
    from nerfactor_amd.nerfactor import geometry_from_nerf as G
    from nerfactor_amd.nerfactor.config import make_config
    cfg = make_config('nerf')
    model = make('nerf', cuda)
    for pref, net in zip(('coarse_', 'fine_'), gi.trained_nerf_nets()):
        for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
            fill(model.net[pref + part], net[part])
    rayo, rayd, _ = gi.nerf1k_rays()
    rayo, rayd = rayo[gi.GEO1K_RAYS], nerf_ref.l2_normalize(rayd[gi.GEO1K_RAYS], 1, 1e-12)
    with torch.no_grad():
        occu, depth, normal = (t.cpu().numpy() for t in
                               G.compute_depth_and_normal(model, dev(rayo, cuda), dev(rayd, cuda), cfg))
    _excluded(np.abs(occu - GOLD['geo1k_occu']), 3e-2, 'geometry occu')
    _excluded(np.abs(depth - GOLD['geo1k_depth']), 0.12, 'geometry depth')
    # expected normal = sum_i w_i n_i of per-sample UNIT density gradients.  After 700 training steps the field's
    # gradient is still noisy: the per-sample normals largely cancel (|sum| = 0.01 .. 0.25 on the surface rays), so the
    # DIRECTION of the sum is ill-conditioned while the vector itself is not — the bound is absolute, on the vector
    want_n = GOLD['geo1k_normal']
    err_n = np.abs(normal - want_n).max(1)
    mag = np.linalg.norm(want_n, axis=1)
    print("geometry normal: |reference| quantiles %s, abs error quantiles (50/90/98/100 %%) %s" % (
        np.round(np.quantile(mag, [.1, .5, .9]), 3), np.round(np.quantile(err_n, [.5, .9, .98, 1.]), 4)))
    _excluded(err_n, 8e-2, 'geometry normal (vector)')
    with torch.no_grad():
        lvis = G.compute_light_visibility(model, dev(GOLD['geo1k_surf'], cuda), dev(GOLD['geo1k_surf_normal'], cuda), cfg,
                                          lvis_far=1., light_h=gi.GEOM_LIGHT_H).cpu().numpy()
    want = GOLD['geo1k_lvis']
    assert np.mean((lvis == 0) != (want == 0)) < 0.01
    _excluded(np.abs(lvis - want).reshape(-1), 4e-2, 'geometry visibility')
