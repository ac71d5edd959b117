"""Generates tests/golden/reference_models.npz by running the REAL google/nerfactor model classes.

    python tests/golden/make_reference_golden.py        (build container only: needs /root/reference)

TensorFlow is not installable here, so the reference's Python runs on tests/golden/tf_shim — a NumPy implementation
of the TensorFlow calls it makes (see tf_shim/README.md).  The model code itself is imported from /root/reference
unmodified and configured from the reference's own nerfactor/config/*.ini files (only paths are overridden):

  nerfactor/models/nerf.py                 Model.call (coarse + fine render), gen_z_fine, accumulate_sigma, compute_loss
  nerfactor/models/shape.py                Model.call (normal + light-visibility MLPs), compute_loss
  nerfactor/models/brdf.py                 Model._eval_brdf_at (learned BRDF MLP, both reciprocal halves), compute_loss
  nerfactor/models/nerfactor.py            Model.call (test, OLAT relight) / train-mode call with jitter + compute_loss
  nerfactor/models/nerfactor_microfacet.py Model.call + compute_loss
  nerfactor/geometry_from_nerf.py          compute_depth_and_normal (GradientTape normals), compute_light_visibility
  nerfactor/datasets/nerf.py, nerf_shape.py  Dataset._glob / _load_data / _process_example_postcache (vali, test)
  nerfactor/util/geom.py                   gen_world2local, dir2rusink
  brdf/microfacet/microfacet.py            Microfacet.__call__

Weights are not stored: they come from the deterministic generators in oracle/ and tests/common.py (seeds below),
which the parity tests call again; a checksum of every weight set is stored so generator drift is detected.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('NERFACTOR_REFERENCE', '/root/reference')
sys.path[:0] = [os.path.join(HERE, 'tf_shim'), REF, os.path.join(REF, 'nerfactor'), REPO]

import tensorflow as tf  # noqa: E402  (the shim)

assert 'numpy-shim' in tf.__version__
from nerfactor.util import io as ioutil  # noqa: E402
from nerfactor.util import geom as geomutil  # noqa: E402
from nerfactor.models.nerf import Model as NerfModel  # noqa: E402
from nerfactor.models.shape import Model as ShapeModel  # noqa: E402
from nerfactor.models.brdf import Model as BrdfModel  # noqa: E402
from nerfactor.models.nerfactor import Model as NerfactorModel  # noqa: E402
from nerfactor.models.nerfactor_microfacet import Model as MicrofacetModel  # noqa: E402
from brdf.microfacet.microfacet import Microfacet  # noqa: E402
from nerfactor import geometry_from_nerf as gfn  # noqa: E402
from nerfactor.datasets.nerf import Dataset as NerfDataset  # noqa: E402
from nerfactor.datasets.nerf_shape import Dataset as ShapeDataset  # noqa: E402

from oracle import nerf_ref, nerfactor_ref  # noqa: E402  (weight generators only)
from tests import common  # noqa: E402
from tests import synth_scene  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402

OUT = {}


def put(key, value):
    a = np.asarray(value)
    assert a.dtype != np.float64, (key, 'float64 leaked out of the fp32 model code')
    assert np.all(np.isfinite(a)) if a.dtype.kind == 'f' else True, key
    OUT[key] = a


def set_layers(network, pairs):
    assert len(network.layers) == len(pairs), (len(network.layers), len(pairs))
    for layer, (k, b) in zip(network.layers, pairs):
        layer.set_weights([k, b])


def ref_config(name, **override):
    cfg = ioutil.read_config(os.path.join(REF, 'nerfactor', 'config', name))
    for k, v in override.items():
        cfg.set('DEFAULT', k, str(v))
    return cfg


def as_tensors(*arrays):
    return tuple(tf.convert_to_tensor(a) for a in arrays)


# ------------------------------------------------------------------------------------------------ NeRF
def run_nerf():
    cfg = ref_config('nerf.ini')
    model = NerfModel(cfg)
    nets = common.nerf_nets(seed=gi.NERF_SEED)
    for pref, net in zip(('coarse_', 'fine_'), nets):
        for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
            set_layers(model.net[pref + part], net[part])
    put('nerf_weight_checksum', gi.checksum_nerf(nets))
    rayo, rayd, gt = gi.nerf_rays()
    n = rayo.shape[0]
    batch = (np.array([b'x'] * n), np.tile(np.int32([[4, n // 4]]), (n, 1))) + as_tensors(rayo, rayd, gt)
    pred, gt_t, loss_kwargs, to_vis = model.call(batch, mode='test')
    for lvl in ('coarse', 'fine'):
        for k in ('rgb', 'occu', 'depth', 'disp'):
            put('nerf_%s_%s' % (lvl, k), to_vis['%s_%s' % (lvl, k)])
    put('nerf_loss', model.compute_loss(pred, gt_t, **loss_kwargs))
    # stage-level anchors: the fine sampler and the compositing weights on inputs that do not depend on an MLP
    z, w, sigma, rd = gi.sampler_inputs()
    put('nerf_z_fine', NerfModel.gen_z_fine(tf.convert_to_tensor(z), tf.convert_to_tensor(w), 128, perturb=False))
    put('nerf_acc_weights', NerfModel.accumulate_sigma(*as_tensors(sigma, z, rd)))
    put('nerf_gen_z_disp', NerfModel.gen_z(2., 6., 64, 3, lin_in_disp=True))
    put('nerf_gen_z', NerfModel.gen_z(2., 6., 64, 3))
    return model, cfg


def run_nerf_trained():
    """The larger fixture: 1024 rays of a 32 x 32 view through the TRAINED networks (tests/golden/nerf_trained_fp16.npz)."""
    cfg = ref_config('nerf.ini')
    model = NerfModel(cfg)
    nets = gi.trained_nerf_nets()
    for pref, net in zip(('coarse_', 'fine_'), nets):
        for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out'):
            set_layers(model.net[pref + part], net[part])
    put('nerf1k_weight_checksum', gi.checksum_nerf(nets))
    rayo, rayd, gt = gi.nerf1k_rays()
    n = rayo.shape[0]
    batch = (np.array([b'x'] * n), np.tile(np.int32([gi.NERF1K_HW]), (n, 1))) + as_tensors(rayo, rayd, gt)
    pred, gt_t, loss_kwargs, to_vis = model.call(batch, mode='test')
    for lvl in ('coarse', 'fine'):
        for k in ('rgb', 'occu', 'depth'):
            put('nerf1k_%s_%s' % (lvl, k), to_vis['%s_%s' % (lvl, k)])
    put('nerf1k_loss', model.compute_loss(pred, gt_t, **loss_kwargs))
    # geometry_from_nerf on the same trained field: a real surface (the unit sphere), 256 rays + 24 x 128 shadow rays
    flags = gfn.FLAGS
    flags.light_h, flags.lpix_chunk, flags.mlp_chunk, flags.lvis_far, flags.scene_bbox = gi.GEOM_LIGHT_H, 8, 65536, 1., None
    ro, rd = as_tensors(rayo[gi.GEO1K_RAYS], rayd[gi.GEO1K_RAYS])
    rd = tf.linalg.l2_normalize(rd, axis=1)
    occu, depth, normal = gfn.compute_depth_and_normal(model, ro, rd, cfg)
    put('geo1k_occu', occu)
    put('geo1k_depth', depth)
    put('geo1k_normal', normal)
    hit = np.flatnonzero(np.asarray(occu) > 0.5)
    idx = hit[np.linspace(0, len(hit) - 1, gi.GEO1K_SURF).astype(int)]
    surf = np.asarray(ro + rd * depth[:, None])[idx]
    nrm = np.asarray(normal)[idx]
    put('geo1k_surf_idx', idx.astype(np.int32))
    put('geo1k_surf', surf)
    put('geo1k_surf_normal', nrm)
    put('geo1k_lvis', gfn.compute_light_visibility(model, *as_tensors(surf, nrm), cfg))


# ------------------------------------------------------------------------------------------------ geometry_from_nerf
def run_geometry(model, cfg):
    """The reference's surface extraction on the same NeRF: 128 + 192 samples per ray, normals from the batch Jacobian
    of the fine density (forward-mode tape of the shim), then shadow rays from 6 surface points to 8 x 16 lights."""
    flags = gfn.FLAGS
    flags.light_h, flags.lpix_chunk, flags.mlp_chunk, flags.lvis_far, flags.scene_bbox = gi.GEOM_LIGHT_H, 5, 4096, 1., None
    rayo, rayd, _ = gi.nerf_rays()
    rayo, rayd = as_tensors(rayo[:gi.GEOM_RAYS], rayd[:gi.GEOM_RAYS])
    rayd = tf.linalg.l2_normalize(rayd, axis=1)
    occu, depth, normal = gfn.compute_depth_and_normal(model, rayo, rayd, cfg)
    put('geo_occu', occu)
    put('geo_depth', depth)
    put('geo_normal', normal)
    surf = np.asarray(rayo + rayd * depth[:, None])[gi.GEOM_SURF_IDX]
    nrm = np.asarray(normal)[gi.GEOM_SURF_IDX]
    put('geo_surf', surf)
    put('geo_surf_normal', nrm)
    put('geo_lvis', gfn.compute_light_visibility(model, *as_tensors(surf, nrm), cfg))
    flags.scene_bbox = gi.GEOM_BBOX
    occu, depth, normal = gfn.compute_depth_and_normal(model, rayo, rayd, cfg)
    put('geo_bbox_occu', occu)
    put('geo_bbox_depth', depth)
    put('geo_bbox_normal', normal)
    put('geo_bbox_lvis', gfn.compute_light_visibility(model, *as_tensors(*gi.geom_bbox_points()), cfg))
    flags.scene_bbox = None


# ------------------------------------------------------------------------------------------------ geometry helpers
def run_geom():
    nrm, a, b = gi.frame_inputs()
    put('geom_world2local', geomutil.gen_world2local(tf.convert_to_tensor(nrm)))
    put('geom_rusink', geomutil.dir2rusink(*as_tensors(a, b)))
    l, v, n, alb, rough = gi.microfacet_inputs()
    put('microfacet_brdf', Microfacet(f0=0.04)(*as_tensors(l, v, n), albedo=tf.convert_to_tensor(alb),
                                               rough=tf.convert_to_tensor(rough)))
    put('microfacet_default', Microfacet()(*as_tensors(l, v, n)))


# ------------------------------------------------------------------------------------------------ shape
def shape_batch(n_lights):
    rayo, rgb, alpha, xyz, normal, lvis = gi.surface_batch(n_lights)
    n = rayo.shape[0]
    id_ = np.array([b'x'] * n)
    hw = np.tile(np.int32([[4, n // 4]]), (n, 1))
    rayd = np.zeros_like(rayo)
    return (id_, hw) + as_tensors(rayo, rayd, rgb, alpha, xyz, normal, lvis)


def run_shape():
    cfg = ref_config('shape.ini', xyz_jitter_std=0)
    model = ShapeModel(cfg)
    net = gi.nerfactor_net(3)
    for part in ('normal_mlp', 'normal_out', 'lvis_mlp', 'lvis_out'):
        set_layers(model.net[part], net[part])
    batch = shape_batch(512)
    pred, gt, loss_kwargs, _ = model.call(batch, mode='test')
    put('shape_normal', pred['normal'])
    put('shape_lvis', pred['lvis'])
    put('shape_loss', model.compute_loss(pred, gt, **loss_kwargs))
    put('shape_lxyz', model.lxyz)


# ------------------------------------------------------------------------------------------------ BRDF prior
def brdf_workdir(tmp):
    data_root = os.path.join(tmp, 'merl_npz')
    os.makedirs(data_root)
    for name in gi.BRDF_NAMES:
        open(os.path.join(data_root, 'train_%s.npz' % name), 'wb').close()
    return data_root


def run_brdf(tmp):
    cfg = ref_config('brdf.ini', data_root=brdf_workdir(tmp))
    model = BrdfModel(cfg)
    assert model.brdf_names == sorted(gi.BRDF_NAMES)
    bnet = gi.brdf_net()
    set_layers(model.net['brdf_mlp'], bnet['brdf_mlp'])
    set_layers(model.net['brdf_out'], bnet['brdf_out'])
    model.latent_code.z = gi.latent_codes()
    i, rusink, refl = gi.brdf_batch()
    n = rusink.shape[0]
    batch = (np.array([b'x'] * n), tf.convert_to_tensor(i), None, None, None) + as_tensors(rusink, refl)
    pred, gt, loss_kwargs, to_vis = model.call(batch, mode='vali')
    put('brdf_pred', pred['brdf'])
    put('brdf_pred_reci', pred['brdf_reci'])
    put('brdf_z', to_vis['z'])
    put('brdf_loss', model.compute_loss(pred, gt, **loss_kwargs))
    put('brdf_interp', model.latent_code.interp(0.25, 0, 0.75, 2))
    return model


# ------------------------------------------------------------------------------------------------ NeRFactor
def nerfactor_workdir(tmp, brdf_root):
    """The directory layout get_config_ini() expects: <root>/<xname>.ini next to <root>/<xname>/checkpoints/ckpt-N."""
    paths = {}
    for name, ini, over in (('shape', 'shape.ini', {}), ('brdf', 'brdf.ini', {'data_root': brdf_root})):
        root = os.path.join(tmp, name)
        os.makedirs(os.path.join(root, 'lr1e-2', 'checkpoints'))
        ioutil.write_config(ref_config(ini, **over), os.path.join(root, 'lr1e-2.ini'))
        paths[name] = os.path.join(root, 'lr1e-2', 'checkpoints', 'ckpt-1')
    envdir = os.path.join(tmp, 'envmaps')
    os.makedirs(envdir)
    return paths, envdir


class RecordNormal:
    """Wraps tf.random.normal so the jitter the reference draws can be replayed by the parity test."""
    def __init__(self):
        self.draws = []
        self.orig = tf.random.normal

    def __call__(self, shape, **kw):
        x = self.orig(shape, **kw)
        self.draws.append(np.asarray(x))
        return x


def run_nerfactor_256(tmp, brdf_root, learned):
    """The larger surface fixture: 256 points x 512 lights in test mode (no jitter); light visibility stored for every
    8th light."""
    paths, envdir = nerfactor_workdir(tmp, brdf_root)
    tag = 'nfl256' if learned else 'nfm256'
    ini = 'nerfactor.ini' if learned else 'nerfactor_microfacet.ini'
    over = dict(shape_model_ckpt=paths['shape'], test_envmap_dir=envdir, embed_light_h=16)
    if learned:
        over['brdf_model_ckpt'] = paths['brdf']
    model = (NerfactorModel if learned else MicrofacetModel)(ref_config(ini, **over), debug=True)
    net = gi.nerfactor_net(3 if learned else 1)
    for part in net:
        set_layers(model.net[part], net[part])
    if learned:
        bnet = gi.brdf_net()
        set_layers(model.brdf_model.net['brdf_mlp'], bnet['brdf_mlp'])
        set_layers(model.brdf_model.net['brdf_out'], bnet['brdf_out'])
    model._light = tf.Variable(gi.light_probe(gi.LIGHT_SCALE['nfl' if learned else 'nfm']))
    rayo, rgb, alpha, xyz, normal, lvis = gi.surface_batch(512, n=gi.SURF256, seed=37)
    n = rayo.shape[0]
    batch = (np.array([b'x'] * n), np.tile(np.int32([[16, n // 16]]), (n, 1))) + as_tensors(
        rayo, np.zeros_like(rayo), rgb, alpha, xyz, normal, lvis)
    pred, gt, loss_kwargs, _ = model.call(batch, mode='test')
    for k in ('rgb', 'normal', 'albedo', 'brdf'):
        put('%s_%s' % (tag, k), pred[k])
    put('%s_lvis' % tag, np.asarray(pred['lvis'])[:, ::gi.LVIS_STRIDE])
    loss_kwargs['mode'] = 'vali'
    put('%s_vali_loss' % tag, model.compute_loss(pred, gt, **loss_kwargs))


def run_nerfactor(tmp, brdf_root, learned):
    paths, envdir = nerfactor_workdir(tmp, brdf_root)
    tag = 'nfl' if learned else 'nfm'
    ini = 'nerfactor.ini' if learned else 'nerfactor_microfacet.ini'
    over = dict(shape_model_ckpt=paths['shape'], test_envmap_dir=envdir, embed_light_h=16, light_tv_weight=2e-4,
                light_achro_weight=1e-4)
    if learned:
        over['brdf_model_ckpt'] = paths['brdf']
    cls = NerfactorModel if learned else MicrofacetModel
    model = cls(ref_config(ini, **over), debug=True)        # debug: 2 x 2 OLAT lights instead of 16 x 32
    z_dim = 3 if learned else 1
    net = gi.nerfactor_net(z_dim)
    for part in net:
        set_layers(model.net[part], net[part])
    if learned:
        bnet = gi.brdf_net()
        set_layers(model.brdf_model.net['brdf_mlp'], bnet['brdf_mlp'])
        set_layers(model.brdf_model.net['brdf_out'], bnet['brdf_out'])
    model._light = tf.Variable(gi.light_probe(gi.LIGHT_SCALE[tag]))
    batch = shape_batch(512)
    # test mode with OLAT relighting
    pred, gt, loss_kwargs, _ = model.call(batch, mode='test', relight_olat=True)
    for k in ('rgb', 'normal', 'lvis', 'albedo', 'brdf', 'rgb_olat'):
        put('%s_test_%s' % (tag, k), pred[k])
    loss_kwargs['mode'] = 'vali'
    put('%s_vali_loss' % tag, model.compute_loss(pred, gt, **loss_kwargs))
    # train mode: jittered second evaluation + smoothness terms + light priors
    rec = RecordNormal()
    tf.random.normal = rec
    try:
        pred, gt, loss_kwargs, _ = model.call(batch, mode='train')
    finally:
        tf.random.normal = rec.orig
    assert len(rec.draws) == 1
    put('%s_train_jitter' % tag, rec.draws[0])
    put('%s_train_rgb' % tag, pred['rgb'])
    put('%s_train_loss' % tag, model.compute_loss(pred, gt, **loss_kwargs))
    put('%s_olat_keys' % tag, np.array(list(model.novel_olat.keys())))
    # the editing hooks of Model.call (used by nerfactor/test.py): albedo scaling / override, BRDF override
    scales, ao_flat, ao_map, z_over = gi.edit_inputs(z_dim)
    for name, kw in (('scaled', dict(albedo_scales=tf.convert_to_tensor(scales), brdf_z_override=tf.convert_to_tensor(z_over))),
                     ('flat_albedo', dict(albedo_override=tf.convert_to_tensor(ao_flat))),
                     ('albedo_map', dict(albedo_override=tf.convert_to_tensor(ao_map)))):
        pred, _, _, _ = model.call(batch, mode='test', **kw)
        put('%s_edit_%s_rgb' % (tag, name), pred['rgb'])
        put('%s_edit_%s_albedo' % (tag, name), pred['albedo'])
        put('%s_edit_%s_brdf' % (tag, name), pred['brdf'])
    if not learned:
        # shape_mode = nerf: no shape MLPs, the NeRF-derived normals and visibility of the batch are used as they are
        over_nerf = dict(over, shape_mode='nerf')
        model = cls(ref_config(ini, **over_nerf), debug=True)
        for part in ('albedo_mlp', 'albedo_out', 'brdf_z_mlp', 'brdf_z_out'):
            set_layers(model.net[part], net[part])
        model._light = tf.Variable(gi.light_probe(gi.LIGHT_SCALE[tag]))
        pred, gt, loss_kwargs, _ = model.call(batch, mode='test')
        for k in ('rgb', 'normal', 'lvis'):
            put('%s_shapenerf_%s' % (tag, k), pred[k])


# ------------------------------------------------------------------------------------------------ datasets
def run_datasets(tmp):
    """The reference's dataset classes on a synthetic scene in its on-disk layout (tests/synth_scene.py, 12 x 16 views):
    ray generation from metadata.json, RGBA compositing, the NeRF-derived buffers, flattening to per-ray batches."""
    data_root, nerf_root = synth_scene.write_scene(tmp, **gi.SCENE_KW)
    over = dict(data_root=data_root, data_nerf_root=nerf_root, imh=gi.SCENE_KW['imh'])

    def load(ds):       # _process_example_precache: tf.py_function(_load_data, Tout=(string, float32, ...)) casts
        id_, *arrays = ds._load_data(ds.files[0])
        return (id_,) + tuple(tf.convert_to_tensor(np.asarray(a, np.float32)) for a in arrays)

    for mode in ('vali', 'test'):
        ds = NerfDataset(ref_config('nerf.ini', **over), mode)
        batch = ds._process_example_postcache(*load(ds))
        assert str(np.asarray(batch[0])[0]) == ('val_000' if mode == 'vali' else 'test_000')
        put('ds_nerf_%s_hw' % mode, batch[1])
        for k, v in zip(('rayo', 'rayd', 'rgb'), batch[2:]):
            put('ds_nerf_%s_%s' % (mode, k), v)
        ds = ShapeDataset(ref_config('nerfactor.ini', **over), mode)
        batch = ds._process_example_postcache(*load(ds))
        for k, v in zip(('rayo', 'rayd', 'rgb', 'alpha', 'xyz', 'normal', 'lvis'), batch[2:]):
            put('ds_shape_%s_%s' % (mode, k), v)
    put('ds_n_train_views', np.int32(len(NerfDataset(ref_config('nerf.ini', **over), 'train').files)))
    ds = NerfDataset(ref_config('nerf.ini', **over), 'test', always_all_rays=True, spp=4)     # 2 x 2 rays per pixel
    batch = ds._process_example_postcache(*load(ds))
    put('ds_nerf_spp4_hw', batch[1])
    put('ds_nerf_spp4_rayd', batch[3])


def main():
    tf.random.set_seed(7)
    model, cfg = run_nerf()
    run_geometry(model, cfg)
    run_nerf_trained()
    run_geom()
    run_shape()
    with tempfile.TemporaryDirectory() as tmp:
        run_datasets(os.path.join(tmp, 'scene'))
        run_brdf(tmp)
        root = os.path.join(tmp, 'merl_npz')
        run_nerfactor(os.path.join(tmp, 'a'), root, learned=True)
        run_nerfactor(os.path.join(tmp, 'b'), root, learned=False)
        run_nerfactor_256(os.path.join(tmp, 'c'), root, learned=True)
        run_nerfactor_256(os.path.join(tmp, 'd'), root, learned=False)
    path = os.path.join(HERE, 'reference_models.npz')
    np.savez_compressed(path, **OUT)
    print('wrote %s (%.1f KiB)' % (path, os.path.getsize(path) / 1024))
    for k, v in OUT.items():
        print('  %-24s %-16s %s' % (k, v.shape, v.dtype))


if __name__ == '__main__':
    main()
