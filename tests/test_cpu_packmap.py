"""The device re-pack maps (ops.DevicePacker) against the host packers, on the CPU: gathering random parameters
through the maps (NumPy stand-in for nfx_pack_gather) must reproduce every host-packed blob bit for bit."""
import numpy as np
import pytest
import torch


def _bf16_bits(x):
    return torch.from_numpy(np.ascontiguousarray(x, np.float32)).to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def _emulate_gather(packer, arrays):
    src = np.concatenate([a.reshape(-1) for a in arrays]).astype(np.float32)
    m = packer.map_host.astype(np.int64)
    take = lambda idx: np.where(idx >= 0, src[np.maximum(idx, 0) & 0x3fffffff], 0.).astype(np.float32)
    fp32 = m[:, 1] == -2

    def half(idx):      # pack_gather.hip: bit 30 of an index = the LO half of the fp32-class pair, bf16(v - bf16(v))
        v = take(idx)
        hi = (_bf16_bits(v).astype(np.uint32) << 16).view(np.float32)
        return np.where((idx >= 0) & ((idx & (1 << 30)) != 0), _bf16_bits(v - hi), _bf16_bits(v))
    pair = half(m[:, 0]).astype(np.uint32) | (half(np.where(fp32, -1, m[:, 1])).astype(np.uint32) << 16)
    return np.where(fp32, take(m[:, 0]).view(np.uint32), pair).astype(np.uint32).view(np.uint8)


def _cases(nfx):
    from nerfactor_amd import ops
    nerf_k = list(ops.NERF_LAYER_SHAPES)

    def m128(ind, out):
        return [(ind, 128), (128, 128), (128, 128), (128 + ind, 128), (128, out)]
    return {
        'nerf': (lambda k, b: ops.pack_nerf_weights(k, b), nerf_k),
        'nerf_train': (lambda k, b: ops.pack_nerf_train_weights(k, b), nerf_k),
        'nerf_geom': (lambda k, b: ops.pack_nerf_geom_weights(k, b), nerf_k),
        'normal': (lambda k, b: ops.pack_mlp128_weights(k, b, nfx.IN_XYZ, 3), m128(63, 3)),
        'lvis': (lambda k, b: ops.pack_mlp128_weights(k, b, nfx.IN_XYZ_LDIR, 1), m128(90, 1)),
        'brdf': (lambda k, b: ops.pack_mlp128_weights(k, b, nfx.IN_Z_RUSINK, 1, z_dim=3), m128(18, 1)),
        'normal_train': (lambda k, b: ops.pack_mlp128_train_weights(k, b, nfx.IN_XYZ, 3), m128(63, 3)),
        'lvis_train': (lambda k, b: ops.pack_mlp128_train_weights(k, b, nfx.IN_XYZ_LDIR, 1), m128(90, 1)),
        'brdf_train': (lambda k, b: ops.pack_brdf_train_weights(k, b, 3), m128(18, 1)),
        # precision = fp32: the split hi / lo fragments of the tuned fp32-class kernels (round 5: gathers of (hi, lo) halves)
        'nerf_fp32': (lambda k, b: ops.pack_nerf_weights(k, b, 'fp32'), nerf_k),
        'nerf_geom_fp32': (lambda k, b: ops.pack_nerf_geom_weights(k, b, 'fp32'), nerf_k),
        'normal_fp32': (lambda k, b: ops.pack_mlp128_weights(k, b, nfx.IN_XYZ, 3, prec='fp32'), m128(63, 3)),
        'lvis_fp32': (lambda k, b: ops.pack_mlp128_weights(k, b, nfx.IN_XYZ_LDIR, 1, prec='fp32'), m128(90, 1)),
        'brdf_fp32': (lambda k, b: ops.pack_mlp128_weights(k, b, nfx.IN_Z_RUSINK, 1, z_dim=3, prec='fp32'), m128(18, 1)),
    }


@pytest.mark.parametrize('name', ['nerf', 'nerf_train', 'nerf_geom', 'normal', 'lvis', 'brdf', 'normal_train', 'lvis_train',
                                  'brdf_train', 'nerf_fp32', 'nerf_geom_fp32', 'normal_fp32', 'lvis_fp32', 'brdf_fp32'])
def test_gather_map_reproduces_host_packer(nfx_lib, name):
    from nerfactor_amd import ops
    pack_fn, shapes_k = _cases(nfx_lib)[name]
    shapes_b = [(s[1],) for s in shapes_k]
    packer = ops.DevicePacker(pack_fn, shapes_k, shapes_b)
    rng = np.random.default_rng(len(name))
    arrays = [rng.normal(size=s).astype(np.float32) for s in shapes_k + shapes_b]
    want = pack_fn(arrays[:len(shapes_k)], arrays[len(shapes_k):]).numpy()
    got = _emulate_gather(packer, arrays)
    assert got.shape == want.shape and packer.nbytes == want.size
    assert np.array_equal(got, want)
    n_kernel = sum(int(np.prod(s)) for s in shapes_k)
    m = packer.map_host
    fp32 = m[:, 1] == -2
    if not name.startswith('nerf_geom'):   # (its fp32 region also carries the sigma_out kernel)
        assert (m[fp32, 0][m[fp32, 0] >= 0] >= n_kernel).all()         # fp32 words gather biases only
    idx = m[~fp32]
    assert (idx[idx >= 0] & 0x3fffffff).max() < n_kernel               # bf16 halves gather kernels only
    assert bool(((idx >= 0) & ((idx & (1 << 30)) != 0)).any()) == name.endswith('_fp32')     # residual halves: fp32-class blobs only


def test_fp32_split_blob_is_hi_lo_of_the_bf16_blob(nfx_lib):
    """NFX_PREC_FP32 blob = [bf16(W) fragments | bf16(W - bf16(W)) fragments | fp32 biases]: a gather of (hi, lo) halves
    (round 5: ops.DevicePacker maps it, test_gather_map_reproduces_host_packer[*_fp32])."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(5)
    ks = [rng.normal(size=s).astype(np.float32) for s in ops.NERF_LAYER_SHAPES]
    bs = [rng.normal(size=(s[1],)).astype(np.float32) for s in ops.NERF_LAYER_SHAPES]
    b16 = ops.pack_nerf_weights(ks, bs, 'bf16').numpy()
    f32 = ops.pack_nerf_weights(ks, bs, 'fp32').numpy()
    nw = 1272 * 1024
    assert f32.size == b16.size + nw
    assert np.array_equal(f32[:nw], b16[:nw]) and np.array_equal(f32[2 * nw:], b16[nw:])
    hi = (b16[:nw].view(np.uint16).astype(np.uint32) << 16).view(np.float32)
    lo = (f32[nw:2 * nw].view(np.uint16).astype(np.uint32) << 16).view(np.float32)
    # every packed weight is reproduced to 2^-16 relative by hi + lo
    ks_lo = [k - (_bf16_bits(k).astype(np.uint32) << 16).view(np.float32).reshape(k.shape) for k in ks]
    want_lo = ops.pack_nerf_weights(ks_lo, bs, 'bf16').numpy()[:nw]
    assert np.array_equal(f32[nw:2 * nw], want_lo)
    assert np.abs(lo).max() <= 2. ** -8 * np.abs(hi).max()


def test_fp32_split_blobs_of_the_width128_networks(nfx_lib):
    """NFX_PREC_FP32 of the surface MLPs (mlp128_x3.hip): [hi | lo | biases] with the bf16 blob's fragment order for
    the xyz heads and the learned BRDF; the light-visibility blob is the plain 90-input network (no per-point fold):
    168 fragments per half."""
    from nerfactor_amd import _capi, ops
    rng = np.random.default_rng(6)
    for kind, in_dims, zd, frags in ((_capi.IN_XYZ, 63, 0, 136), (_capi.IN_Z_RUSINK, 18, 3, 136),
                                     (_capi.IN_XYZ_LDIR, 90, 0, 168)):
        shapes = [(in_dims, 128), (128, 128), (128, 128), (128 + in_dims, 128), (128, 1)]
        ks = [rng.normal(size=s).astype(np.float32) for s in shapes]
        bs = [rng.normal(size=(s[1],)).astype(np.float32) for s in shapes]
        f32 = ops.pack_mlp128_weights(ks, bs, kind, 1, z_dim=zd, prec='fp32').numpy()
        nw = frags * 1024
        assert f32.size == 2 * nw + 544 * 4
        ks_lo = [k - (_bf16_bits(k).astype(np.uint32) << 16).view(np.float32).reshape(k.shape) for k in ks]
        if kind != _capi.IN_XYZ_LDIR:
            b16 = ops.pack_mlp128_weights(ks, bs, kind, 1, z_dim=zd).numpy()
            assert np.array_equal(f32[:nw], b16[:nw]) and np.array_equal(f32[2 * nw:], b16[nw:])
            assert np.array_equal(f32[nw:2 * nw], ops.pack_mlp128_weights(ks_lo, bs, kind, 1, z_dim=zd).numpy()[:nw])
        else:
            # every kernel entry appears exactly once in the hi half (zero padding aside), its residual in the lo half
            hi = (f32[:nw].view(np.uint16).astype(np.uint32) << 16).view(np.float32)
            lo = (f32[nw:2 * nw].view(np.uint16).astype(np.uint32) << 16).view(np.float32)
            want_hi = np.concatenate([(_bf16_bits(k).astype(np.uint32) << 16).view(np.float32).ravel() for k in ks])
            want_lo = np.concatenate([(_bf16_bits(k).astype(np.uint32) << 16).view(np.float32).ravel() for k in ks_lo])
            assert np.array_equal(np.sort(hi[hi != 0]), np.sort(want_hi[want_hi != 0]))
            assert np.array_equal(np.sort(lo[lo != 0]), np.sort(want_lo[want_lo != 0]))
            biases = f32[2 * nw:].view(np.float32)
            assert np.array_equal(biases[:512], np.concatenate(bs[:4])) and biases[512] == bs[4][0]


def test_fp32_split_blob_of_the_density_gradient_kernel(nfx_lib):
    """NFX_PREC_FP32 geometry blob (nerf_geom_x3.hip) = [the bf16 blob's fragments | the same chunk sequence packed from
    W - bf16(W) | the bf16 blob's floats: encoder biases, sigma bias tile, sigma_out kernel in fp32]."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(7)
    ks = [rng.normal(size=s).astype(np.float32) for s in ops.NERF_LAYER_SHAPES]
    bs = [rng.normal(size=(s[1],)).astype(np.float32) for s in ops.NERF_LAYER_SHAPES]
    b16 = ops.pack_nerf_geom_weights(ks, bs).numpy()
    f32 = ops.pack_nerf_geom_weights(ks, bs, 'fp32').numpy()
    n_floats = 8 * 256 + 32 + 256
    nw = b16.size - 4 * n_floats
    assert f32.size == 2 * nw + 4 * n_floats
    assert np.array_equal(f32[:nw], b16[:nw]) and np.array_equal(f32[2 * nw:], b16[nw:])
    ks_lo = [k - (_bf16_bits(k).astype(np.uint32) << 16).view(np.float32).reshape(k.shape) for k in ks]
    assert np.array_equal(f32[nw:2 * nw], ops.pack_nerf_geom_weights(ks_lo, bs).numpy()[:nw])
    assert np.array_equal(f32[2 * nw:].view(np.float32)[-256:], ks[8][:, 0])


@pytest.mark.parametrize('name', ['lvis', 'lvis_train', 'nerf', 'lvis_fp32'])
def test_gather_map_translated_into_a_flat_parameter_buffer(nfx_lib, name):
    """Parameters that are views of one buffer (optim.AMSGrad's flat bucket) are gathered in place: the map's indices
    translated to offsets from the lowest parameter (ops.DevicePacker._map_in_place) reproduce the host-packed blob from
    the buffer itself — whatever the order and the gaps of the views."""
    from nerfactor_amd import ops
    pack_fn, shapes_k = _cases(nfx_lib)[name]
    shapes_b = [(s[1],) for s in shapes_k]
    packer = ops.DevicePacker(pack_fn, shapes_k, shapes_b)
    shapes = shapes_k + shapes_b
    rng = np.random.default_rng(11)
    order = rng.permutation(len(shapes))
    sizes = [int(np.prod(s)) for s in shapes]
    offs, at = {}, 5
    for i in order:                      # shuffled, with gaps (other networks' parameters live in between)
        offs[i] = at
        at += sizes[i] + int(rng.integers(0, 40))
    flat = torch.from_numpy(rng.normal(size=at + 3).astype(np.float32))
    views = [flat[offs[i]:offs[i] + sizes[i]].view(shapes[i]) for i in range(len(shapes))]
    base, dmap = packer._map_in_place(views)
    assert base == min(v.data_ptr() for v in views)
    first = (base - flat.data_ptr()) // 4
    moved = ops.DevicePacker.__new__(ops.DevicePacker)
    moved.map_host = dmap.numpy()
    got = _emulate_gather(moved, [flat.numpy()[first:]])
    want = pack_fn([v.numpy() for v in views[:len(shapes_k)]], [v.numpy() for v in views[len(shapes_k):]]).numpy()
    assert np.array_equal(got, want)
    assert packer._map_in_place(views)[1] is dmap                                  # cached per set of addresses
    assert packer._map_in_place([v.clone() for v in views]) is None                # separate storages: the concatenation path
