"""Deterministic inputs and weights shared by make_reference_golden.py (which feeds them to the reference's model
classes) and the parity tests (which feed them to oracle/ and to the HIP path).  NumPy only; no reference access."""
import numpy as np

from oracle import nerf_ref, nerfactor_ref
from tests import common

NERF_SEED = 21
LIGHT_SCALE = {'nfl': 0.3, 'nfm': 2.}      # keeps most pixels of both variants below the clip at 1
GEOM_RAYS, GEOM_LIGHT_H, GEOM_SURF_IDX = 16, 8, [3, 4, 5, 11, 13, 15]
GEOM_BBOX = '-1.5,1.5,-1.5,1.5,-1.5,1.5'
SCENE_KW = dict(imh=12, imw=16, n_train=2, n_val=1, n_test=1, light_h=4, seed=5)     # tests/synth_scene.write_scene
MVS_SCENE_KW = dict(imh=12, imw=16, n_train=2, n_val=1, n_test=1, light_h=16, seed=3)  # tests/synth_scene.write_mvs_scene
GRAD_NERF_RAYS = 32          # rays of the reference-differentiated NeRF training step (make_reference_grad_golden.py)
BRDF_NAMES = ['alum-bronze', 'blue-fabric', 'chrome', 'delrin', 'nylon']   # sorted, as xm.os.sortglob returns them


def _f32(x):
    return np.ascontiguousarray(x, np.float32)


def checksum(list_of_pairs):
    return np.float32(sum(float(np.abs(k).sum()) + float(np.abs(b).sum()) for k, b in list_of_pairs))


def checksum_nerf(nets):
    return _f32([checksum([p for part in ('enc', 'sigma_out', 'bottleneck', 'rgb_out') for p in net[part]])
                 for net in nets])


def nerf_rays():
    """64 rays of an 8 x 8 camera looking at the origin + random ground-truth colours."""
    rayo, rayd = common.camera_rays(8, 8)
    rng = np.random.default_rng(31)
    return _f32(rayo), _f32(rayd * 1.7), _f32(rng.uniform(0, 1, (rayo.shape[0], 3)))


def sampler_inputs():
    """z [32,64] (perturbed strata), weights [32,64] (peaky; row 0 all-zero, row 1 one-hot), sigma, ray dirs."""
    rng = np.random.default_rng(32)
    z = nerf_ref.gen_z(2., 6., 64, 32, u=rng.uniform(0, 1, (32, 64)).astype(np.float32))
    w = rng.uniform(0, 1, (32, 64)) ** 8
    w[0] = 0
    w[1] = 0
    w[1, 17] = 1
    sigma = rng.normal(0, 4, (32, 64))
    rd = rng.normal(size=(32, 3))
    rd /= np.linalg.norm(rd, axis=1, keepdims=True)
    return _f32(z), _f32(w), _f32(sigma), _f32(rd)


def frame_inputs():
    rng = np.random.default_rng(33)
    nrm = rng.normal(size=(48, 3))
    nrm[0] = (0, 0, 1)
    nrm[1] = (0, 0, -1)
    nrm[2] = (1, 0, 0)
    nrm[3] *= 5
    a = rng.normal(size=(96, 3))
    b = rng.normal(size=(96, 3))
    a[:, 2] = np.abs(a[:, 2]) + 1e-2
    b[:, 2] = np.abs(b[:, 2]) + 1e-2
    b[0] = a[0]                                     # identical directions: theta_d = 0
    return _f32(nrm), _f32(a), _f32(b)


def microfacet_inputs():
    rng = np.random.default_rng(34)
    n_pts, n_l = 12, 40
    l = rng.normal(size=(n_pts, n_l, 3))
    v = rng.normal(size=(n_pts, 3))
    nrm = rng.normal(size=(n_pts, 3))
    v += 2 * nrm / np.linalg.norm(nrm, axis=1, keepdims=True)      # mostly front-facing views
    alb = rng.uniform(0, 1, (n_pts, 3))
    rough = rng.uniform(0.05, 1, (n_pts, 1))
    return _f32(l), _f32(v), _f32(nrm), _f32(alb), _f32(rough)


def trained_nerf_nets():
    """Coarse + fine NeRF networks fitted to the unit-sphere scene (tests/golden/make_trained_nerf_weights.py), stored
    as float16: empty space sits at a robustly negative density, so no ray of the fixtures is decided by the sign of
    a near-zero last-sample logit."""
    import os
    w = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'nerf_trained_fp16.npz'))
    nets = []
    for pref in ('coarse_', 'fine_'):
        net = {}
        for part, n in (('enc', 8), ('sigma_out', 1), ('bottleneck', 1), ('rgb_out', 2)):
            net[part] = [(_f32(w['net_%s%s_layer%d.kernel' % (pref, part, i)]),
                          _f32(w['net_%s%s_layer%d.bias' % (pref, part, i)])) for i in range(n)]
        nets.append(net)
    return nets


NERF1K_HW = (32, 32)            # the 1024-ray view of the larger NeRF fixture
GEO1K_RAYS = slice(384, 640)    # 8 image rows through the middle of that view (geometry fixture, 256 rays)
GEO1K_SURF = 24                 # surface points of the light-visibility fixture on the trained NeRF (x 8 x 16 lights)
SURF256 = 256                   # surface points of the larger NeRFactor fixture
LVIS_STRIDE = 8                 # every 8th light of the 512 is stored for the 256-point fixtures


def nerf1k_rays():
    rayo, rayd = common.camera_rays(*NERF1K_HW, cam_loc=(1.9, -2.8, 2.1))
    rng = np.random.default_rng(36)
    return _f32(rayo), _f32(rayd), _f32(rng.uniform(0, 1, (rayo.shape[0], 3)))


def surface_batch(n_lights, n=24, seed=35):
    """(rayo, rgb, alpha, xyz, normal, lvis) of n surface points, about a fifth of them background (alpha = 0)."""
    rng = np.random.default_rng(seed)
    if n != 24:
        xyz = rng.uniform(-1, 1, (n, 3))
        normal = rng.normal(size=(n, 3))
        normal /= np.linalg.norm(normal, axis=1, keepdims=True)
        rayo = np.tile(np.float32([[2.2, -2.4, 1.9]]), (n, 1))
        rgb = rng.uniform(0, 1, (n, 3))
        alpha = rng.uniform(0.3, 1, (n, 1)) * (rng.uniform(size=(n, 1)) > 0.2)
        lvis = rng.uniform(0, 1, (n, n_lights))
        return _f32(rayo), _f32(rgb), _f32(alpha), _f32(xyz), _f32(normal), _f32(lvis)
    xyz = rng.uniform(-1, 1, (n, 3))
    normal = rng.normal(size=(n, 3))
    normal /= np.linalg.norm(normal, axis=1, keepdims=True)
    rayo = np.tile(np.float32([[2.2, -2.4, 1.9]]), (n, 1))
    rgb = rng.uniform(0, 1, (n, 3))
    alpha = rng.uniform(0.3, 1, (n, 1))
    alpha[[2, 7, 8, 15, 23]] = 0
    lvis = rng.uniform(0, 1, (n, n_lights))
    return _f32(rayo), _f32(rgb), _f32(alpha), _f32(xyz), _f32(normal), _f32(lvis)


def _with_biases(net, seed):
    return nerf_ref.randomize_biases(net, np.random.default_rng(seed))


def nerfactor_net(z_dim):
    return _with_biases(nerfactor_ref.init_nerfactor_net(np.random.default_rng(40 + z_dim), z_dim), 50 + z_dim)


def brdf_net():
    return _with_biases(nerfactor_ref.init_brdf_mlp(np.random.default_rng(44)), 54)


def latent_codes():
    return _f32(np.random.default_rng(45).normal(0, 0.5, (len(BRDF_NAMES), 3)))


def brdf_batch():
    rng = np.random.default_rng(46)
    n = 192
    i = rng.integers(0, len(BRDF_NAMES), n).astype(np.int32)
    rusink = np.stack([rng.uniform(0, np.pi, n), rng.uniform(0, np.pi / 2, n), rng.uniform(0, np.pi / 2, n)], 1)
    refl = np.exp(rng.normal(-2, 2, (n, 1)))
    return i, _f32(rusink), _f32(refl)


def light_probe(scale=2.):
    return _f32(np.random.default_rng(47).uniform(0, 1, (16, 32, 3)) ** 2 * scale)


def geom_bbox_points():
    """Surface points inside GEOM_BBOX whose shadow rays (length 1) partly leave it, and their unit normals."""
    rng = np.random.default_rng(48)
    pts = rng.uniform(-1.2, 1.2, (5, 3))
    nrm = rng.normal(size=(5, 3))
    return _f32(pts), _f32(nrm / np.linalg.norm(nrm, axis=1, keepdims=True))


def edit_inputs(z_dim):
    """(albedo_scales [3], albedo_override [3], albedo_override [24,3], brdf_z_override [z_dim])."""
    rng = np.random.default_rng(49)
    z = rng.uniform(0.2, 0.8, z_dim)
    return _f32([0.9, 1.1, 0.8]), _f32([0.6, 0.3, 0.2]), _f32(rng.uniform(0.1, 0.9, (24, 3))), _f32(z)


def summary(a, n_sample=2048):
    """Compact fingerprint of a large tensor for the gradient fixture: [Frobenius norm, sum, n_sample strided elements]
    (stride = size // n_sample over the flattened array).  Compare with `summary(other)` element by element."""
    flat = np.asarray(a, np.float64).reshape(-1)
    stride = max(1, flat.size // n_sample)
    return np.concatenate(([np.sqrt(np.sum(flat * flat)), flat.sum()], flat[::stride][:n_sample])).astype(np.float32)
