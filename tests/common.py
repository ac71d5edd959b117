"""Shared synthetic-input generators for tests, smoke() and bench.py."""
import numpy as np

from oracle import nerf_ref


def nerf_nets(seed=0, opaque=True, random_bias=True):
    """Coarse + fine NeRF nets, glorot-uniform; 'opaque' scales sigma_out so rays terminate."""
    rng = np.random.default_rng(seed)
    nets = []
    for _ in range(2):
        net = nerf_ref.init_nerf_net(rng, sigma_bias=0.5 if opaque else 0.,
                                     sigma_gain=8. if opaque else 1.)
        if random_bias:
            sb = net['sigma_out'][0][1].copy()
            nerf_ref.randomize_biases(net, rng)
            net['sigma_out'][0] = (net['sigma_out'][0][0], sb + net['sigma_out'][0][1])
        nets.append(net)
    return nets


def nerf_layers(net):
    """The 12 (kernel, bias) pairs in the order of nfx_nerf_pack_weights."""
    layers = list(net['enc']) + [net['sigma_out'][0], net['bottleneck'][0]] + list(net['rgb_out'])
    return [k for k, _ in layers], [b for _, b in layers]


def camera_rays(imh, imw, cam_loc=(2.4, -2.6, 1.8), angle_x=0.6911):
    c2w = nerf_ref.lookat_cam_to_world(np.asarray(cam_loc) * 4. / np.linalg.norm(cam_loc))
    rayo, rayd = nerf_ref.gen_rays(c2w, angle_x, imh, imw)
    return rayo.reshape(-1, 3).astype(np.float32), rayd.reshape(-1, 3).astype(np.float32)


def assert_same_bits(want, got, what, row_len=None):
    """torch.equal with a diagnosis: on a mismatch the message carries the kernel / variant label
    `what`, how many elements and rows differ, the first differing rows, the histogram of the differing elements'
    position modulo 64 / 32 / 16 (a lane-group pattern points at a cross-lane or wait-state hazard, a uniform one at
    data) and the largest difference."""
    import torch
    if torch.equal(want, got):
        return
    assert want.shape == got.shape, (what, tuple(want.shape), tuple(got.shape))
    w, g = want.reshape(-1), got.reshape(-1)
    bad = torch.nonzero((w != g) & ~(torch.isnan(w) & torch.isnan(g))).reshape(-1).cpu()
    row_len = row_len or (want.shape[-1] if want.dim() > 1 else 1)
    rows = torch.unique(bad // row_len)
    cols = bad % row_len
    hist = lambda m: torch.bincount(cols % m, minlength=m).tolist()
    diff = (w[bad.to(w.device)] - g[bad.to(w.device)]).abs()
    raise AssertionError(
        "%s: %d of %d elements differ in %d of %d rows; first rows %s; first elements (row, col, want, got) %s; "
        "column %% 64 histogram %s; column %% 16 histogram %s; max |diff| %.3e" % (
            what, bad.numel(), w.numel(), rows.numel(), w.numel() // row_len, rows[:8].tolist(),
            [(int(i // row_len), int(i % row_len), float(w[i]), float(g[i])) for i in bad[:6]],
            hist(64), hist(16), float(diff.max()) if diff.numel() else 0.))
