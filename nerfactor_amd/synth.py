"""Seeded synthetic inputs of SURVEY.md §8(d) for bench.py and the scripts (product side: NumPy only, nothing from
oracle/ or tests/).  The test-suite has its own generators built on the oracle (tests/common.py);
tests/test_cpu_nerf.py::test_synth_matches_test_generators holds the two to each other.

  C2 (NeRF render)       nerf_nets, nerf_layers, camera_rays
  C3 (NeRFactor render)  surface_batch, probes
"""
import numpy as np

NERF_WIDTH, NERF_DEPTH = 256, 8


def _glorot(rng, fan_in, fan_out):
    """Keras Dense default kernel initialiser (networks/mlp.py:35): U(-l, l), l = sqrt(6 / (in + out))."""
    lim = np.sqrt(6. / (fan_in + fan_out))
    return rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)


def nerf_net(rng, sigma_bias=0., sigma_gain=1., random_bias=True, bias_scale=0.1):
    """One NeRF network with the shapes of models/nerf.py:53-71 as {'enc': [(kernel, bias)] * 8, 'sigma_out',
    'bottleneck', 'rgb_out'}.  Draw order: all kernels first (enc, sigma_out, bottleneck, rgb_out), then the
    biases in the same order."""
    dx, dv, w = 63, 27, NERF_WIDTH
    enc, fan_in = [], dx
    for i in range(NERF_DEPTH):
        enc.append([_glorot(rng, fan_in, w), np.zeros(w, np.float32)])
        fan_in = w + dx if i == NERF_DEPTH // 2 else w
    net = {'enc': enc,
           'sigma_out': [[_glorot(rng, w, 1) * np.float32(sigma_gain), np.full(1, sigma_bias, np.float32)]],
           'bottleneck': [[_glorot(rng, w, w), np.zeros(w, np.float32)]],
           'rgb_out': [[_glorot(rng, w + dv, w // 2), np.zeros(w // 2, np.float32)],
                       [_glorot(rng, w // 2, 3), np.zeros(3, np.float32)]]}
    if random_bias:
        for name, layers in net.items():
            for layer in layers:
                b = rng.uniform(-bias_scale, bias_scale, size=layer[1].shape).astype(np.float32)
                layer[1] = layer[1] + b if name == 'sigma_out' else b
    return {k: [tuple(layer) for layer in v] for k, v in net.items()}


def nerf_nets(seed=0, opaque=True, random_bias=True):
    """Coarse + fine networks; 'opaque' = sigma_out.kernel x 8, sigma_out.bias + 0.5 so that rays terminate."""
    rng = np.random.default_rng(seed)
    return [nerf_net(rng, 0.5 if opaque else 0., 8. if opaque else 1., random_bias) for _ in range(2)]


def nerf_layers(net):
    """The 12 (kernel, bias) pairs in the order of nfx_nerf_pack_weights."""
    layers = list(net['enc']) + [net['sigma_out'][0], net['bottleneck'][0]] + list(net['rgb_out'])
    return [k for k, _ in layers], [b for _, b in layers]


def lookat(cam_loc, target=(0., 0., 0.), up=(0., 0., 1.)):
    """Camera-to-world of a Blender-convention camera (looks down -z, +y up), the layout of the NeRF-synthetic
    `cam_transform_mat`."""
    cam_loc = np.asarray(cam_loc, np.float64)
    fwd = np.asarray(target, np.float64) - cam_loc
    fwd /= np.linalg.norm(fwd)
    right = np.cross(fwd, np.asarray(up, np.float64))
    right /= np.linalg.norm(right)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = right, np.cross(right, fwd), -fwd, cam_loc
    return m


def camera_rays(imh, imw, cam_loc=(2.4, -2.6, 1.8), angle_x=0.6911):
    """Rays of one view (camera on the radius-4 sphere) through the product's own ray generator
    (nerfactor_amd/nerfactor/datasets/nerf.py:_gen_rays, reference datasets/nerf.py:172-193)."""
    from .nerfactor.datasets.nerf import gen_rays
    c2w = lookat(np.asarray(cam_loc) * 4. / np.linalg.norm(cam_loc))
    rayo, rayd = gen_rays(c2w, angle_x, imh, imw)
    return rayo.reshape(-1, 3).astype(np.float32), rayd.reshape(-1, 3).astype(np.float32)


def surface_batch(n, seed=1, fg_frac=0.6, n_lights=512, cam=(2.2, -2.4, 1.7)):
    """NeRFactor batch tuple of SURVEY §8d "C3": n surface points in [-1, 1]^3, unit normals, Bernoulli(fg_frac)
    alpha, uniform visibility / rgb ground truth.  NumPy arrays in the dataset's order
    (id, hw, rayo, rayd, rgb, alpha, xyz, normal, lvis) with id = hw = None."""
    rng = np.random.default_rng(seed)
    xyz = rng.uniform(-1, 1, size=(n, 3)).astype(np.float32)
    nrm = rng.normal(size=(n, 3)).astype(np.float32)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-6)
    alpha = (rng.uniform(size=(n, 1)) < fg_frac).astype(np.float32)
    rayo = np.broadcast_to(np.asarray(cam, np.float32), (n, 3)).copy()
    rgb = rng.uniform(size=(n, 3)).astype(np.float32)
    lvis = rng.uniform(size=(n, n_lights)).astype(np.float32)
    return (None, None, rayo, np.zeros((n, 3), np.float32), rgb, alpha, xyz, nrm, lvis)


def probes(n, light_h=16, seed=20):
    """n HDR-like light probes [light_h, 2 light_h, 3] ~ exp(N(0, 1)) (SURVEY §8d "C5")."""
    return [np.exp(np.random.default_rng(seed + i).normal(size=(light_h, 2 * light_h, 3))).astype(np.float32)
            for i in range(n)]
