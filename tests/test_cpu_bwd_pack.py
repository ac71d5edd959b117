"""Backward-side weight blobs (NeRF / surface-MLP / learned-BRDF training blobs, density-gradient chain) checked on
the CPU: the host packers' output is pushed through the lane-level MFMA emulation of tests/emu.py and compared with
plain W^T products and, for the forward halves, with the reference dataflow."""
import numpy as np
import pytest

from oracle.nerf_ref import bf16_round
from tests import common, emu

FWD_FRAGS = 1272


def _reader(blob, weight_bytes, first_frag):
    rd = emu.BlobReader(np.asarray(blob), weight_bytes)
    rd.pos = first_frag
    return rd


def _close(got, want):
    np.testing.assert_allclose(got, want, atol=1e-4, rtol=1e-4)


def test_train_blob_dgrad_chain(nfx_lib):
    from nerfactor_amd import ops
    rng = np.random.default_rng(11)
    net = common.nerf_nets(seed=9)[0]
    ks, bs = common.nerf_layers(net)
    blob = ops.pack_nerf_train_weights(ks, bs, 'bf16').numpy()
    wbytes = (FWD_FRAGS + 1136) * 1024
    assert blob.nbytes == nfx_lib.lib.nfx_nerf_train_packed_bytes(0) == wbytes + 2496 * 4
    fwd = ops.pack_nerf_weights(ks, bs, 'bf16').numpy()
    assert np.array_equal(blob[:FWD_FRAGS * 1024], fwd[:FWD_FRAGS * 1024])      # forward fragments unchanged
    assert np.array_equal(blob[wbytes:], fwd[FWD_FRAGS * 1024:])                # biases moved behind the dgrad part
    K = [bf16_round(np.asarray(k, np.float32)).astype(np.float64) for k in ks]
    rd = _reader(blob, wbytes, FWD_FRAGS)
    dz = lambda f: bf16_round(rng.normal(size=(32, f)).astype(np.float32))
    # D1: rgb_out[1]^T (3 of 16 slots used)
    g = np.zeros((32, 16), np.float32)
    g[:, :3] = dz(3)
    _close(emu.dgrad_layer(rd, 4, emu.hidden_ops(g), 4), g[:, :3] @ K[11].T)
    # D2: the feature rows of rgb_out[0]
    g = dz(128)
    _close(emu.dgrad_layer(rd, 8, emu.hidden_ops(g), 8), g @ K[10][:256].T)
    # D3: [bottleneck | sigma_out]^T
    g, gs = dz(256), np.zeros((32, 16), np.float32)
    gs[:, :1] = dz(1)
    _close(emu.dgrad_layer(rd, 20, emu.hidden_ops(g) + emu.hidden_ops(gs), 8), g @ K[9].T + gs[:, :1] @ K[8].T)
    for l in range(7, 0, -1):
        g = dz(256)
        _close(emu.dgrad_layer(rd, 16, emu.hidden_ops(g), 8), g @ K[l][:256].T)
    assert rd.pos == FWD_FRAGS + 1136


def test_geom_blob_backward_chain_and_constants(nfx_lib):
    from nerfactor_amd import ops
    rng = np.random.default_rng(12)
    net = common.nerf_nets(seed=10)[0]
    ks, bs = common.nerf_layers(net)
    blob = ops.pack_nerf_geom_weights(ks, bs, 'bf16').numpy()
    enc_frags = 8 * 8 + 6 * 8 * 16 + 8 * 24             # enc[0] K=64, six K=256 layers, enc[5] K=256+64+pad
    fwd = ops.pack_nerf_weights(ks, bs, 'bf16').numpy()
    geo_fwd = enc_frags + 16
    wbytes = (geo_fwd + (7 * 8 + 4) * 16) * 1024
    assert blob.nbytes == nfx_lib.lib.nfx_nerf_geom_packed_bytes(0) == wbytes + (2048 + 32 + 256) * 4
    assert np.array_equal(blob[:enc_frags * 1024], fwd[:enc_frags * 1024])
    fl = np.frombuffer(blob[wbytes:].tobytes(), np.float32)
    np.testing.assert_array_equal(fl[:2048], np.concatenate([np.asarray(b, np.float32) for b in bs[:8]]))
    assert fl[2048] == np.float32(np.asarray(bs[8]).ravel()[0])
    np.testing.assert_array_equal(fl[2080:], np.asarray(ks[8], np.float32).ravel())   # raw fp32 sigma kernel
    K = [bf16_round(np.asarray(k, np.float32)).astype(np.float64) for k in ks]
    rows = emu.posenc10_slot_rows()
    rd = _reader(blob, wbytes, geo_fwd)
    dz = lambda: bf16_round(rng.normal(size=(32, 256)).astype(np.float32))

    def igrad(kernel_rows):
        g = dz()
        got = emu.dgrad_layer(rd, 16, emu.hidden_ops(g), 2)          # [32 points, 64]: tile tt, C/D feature f
        for tt in range(2):
            for r in range(16):
                for h in range(2):
                    q = 8 * (2 * tt + (r >> 3)) + (r & 7)
                    f = 32 * tt + (r & 3) + 8 * (r >> 2) + 4 * h
                    src = rows[h, q]
                    want = g @ kernel_rows[src] if src >= 0 else np.zeros(32)
                    _close(got[:, f], want)

    for l in (7, 6):
        g = dz()
        _close(emu.dgrad_layer(rd, 16, emu.hidden_ops(g), 8), g @ K[l].T)
    igrad(K[5][256:])
    for l in range(5, 0, -1):
        g = dz()
        _close(emu.dgrad_layer(rd, 16, emu.hidden_ops(g), 8), g @ K[l][:256].T)
    igrad(K[0])
    assert rd.pos * 1024 == wbytes


# ----------------------------------------------------------------------------------------- width-128 training blobs
def _net128(seed, in_dims, out_dims):
    from oracle import nerfactor_ref
    rng = np.random.default_rng(seed)
    layers, out = nerfactor_ref.init_mlp128(rng, in_dims, out_dims)
    pairs = [(k, rng.uniform(-.2, .2, b.shape).astype(np.float32)) for k, b in layers + out]
    return [k for k, _ in pairs], [b for _, b in pairs]


def _mlp128_quant(x, ks, bs):
    """Reference dataflow of mlp.Network([128]*4, skip_at=[2]) + out layer with bf16 operands, fp32 accumulation."""
    q = lambda a: bf16_round(np.asarray(a, np.float32)).astype(np.float64)
    h, x = q(x), q(x)
    for l in range(4):
        h = np.maximum(h @ q(ks[l]) + bs[l], 0)
        h = q(h)
        if l == 2:
            h = np.concatenate((h, x), 1)
    return h @ q(ks[4]) + bs[4]


@pytest.mark.parametrize('lv', [False, True])
def test_mlp128_train_blob_forward_and_dgrad(nfx_lib, lv):
    from nerfactor_amd import ops
    from oracle import nerf_ref
    in_kind = nfx_lib.IN_XYZ_LDIR if lv else nfx_lib.IN_XYZ
    ind, out_dim = (90 if lv else 63), 3
    ks, bs = _net128(21 + lv, ind, out_dim)
    blob = ops.pack_mlp128_train_weights(ks, bs, in_kind, out_dim).numpy()
    p0, p3 = (8, 16) if lv else (4, 12)
    n_frags = 4 * p0 + 4 * 8 + 4 * 8 + 4 * p3 + 8 + 4 * 4 + 3 * 4 * 8
    wbytes = n_frags * 1024
    assert blob.nbytes == nfx_lib.lib.nfx_mlp128_train_packed_bytes(in_kind) == wbytes + 544 * 4
    rng = np.random.default_rng(5)
    pts = rng.uniform(-1, 1, (32, 3)).astype(np.float32)
    ldir = nerf_ref.l2_normalize(rng.normal(size=(32, 3)).astype(np.float32), 1, 1e-12)
    rd = emu.BlobReader(blob, wbytes)
    x_ops = emu.posenc_slots(pts, 10) + (emu.posenc_slots(ldir, 4) if lv else [])
    h = emu.layer(rd, p0, rd.b, 0, x_ops, 4, True)
    h = emu.layer(rd, 8, rd.b, 128, h, 4, True)
    h = emu.layer(rd, 8, rd.b, 256, h, 4, True)
    h = emu.layer(rd, p3, rd.b, 384, h + x_ops, 4, True)
    acc = emu.tile(rd, 8, rd.b, 512, h)
    got = np.stack([acc[:32, r] for r in range(out_dim)], 1)
    x = np.concatenate((nerf_ref.embed(pts, 10),) + ((nerf_ref.embed(ldir, 4),) if lv else ()), 1)
    np.testing.assert_allclose(got, _mlp128_quant(x, ks, bs), atol=3e-3, rtol=3e-3)
    # dgrad chain: out^T (16 gradient slots, out_dim used), then the hidden rows of L3, L2, L1 transposed
    K = [bf16_round(np.asarray(k, np.float32)).astype(np.float64) for k in ks]
    dz = lambda f: bf16_round(rng.normal(size=(32, f)).astype(np.float32))
    g = np.zeros((32, 16), np.float32)
    g[:, :out_dim] = dz(out_dim)
    _close(emu.dgrad_layer(rd, 4, emu.hidden_ops(g), 4), g[:, :out_dim] @ K[4].T)
    for l in (3, 2, 1):
        g = dz(128)
        _close(emu.dgrad_layer(rd, 8, emu.hidden_ops(g), 4), g @ K[l][:128].T)
    assert rd.pos == n_frags


def _brdf_slot_rows(zd):
    """Keras input row of [z, posenc2(rusink)] behind slot q (0..15) of lane half h, or -1 (brdf_tile's layout)."""
    rows = -np.ones((2, 16), np.int64)
    for q in range(6):
        rows[0, q] = zd + 3 + 6 * (q // 3) + (q % 3)
        rows[1, q] = zd + 3 + 6 * (q // 3) + 3 + (q % 3)
    rows[0, 6], rows[1, 6], rows[0, 7], rows[1, 7] = zd, zd + 2, zd + 1, 0
    for j in range(8):
        for hh in range(2):
            if 1 + 2 * j + hh < zd:
                rows[hh, 8 + j] = 1 + 2 * j + hh
    return rows


@pytest.mark.parametrize('zd', [1, 3])
def test_brdf_train_blob_forward_dgrad_and_input_gradient(nfx_lib, zd):
    from nerfactor_amd import ops
    ks, bs = _net128(31 + zd, zd + 15, 1)
    blob = ops.pack_brdf_train_weights(ks, bs, zd).numpy()
    n_frags = 136 + 4 * 4 + 8 + 3 * 4 * 8 + 8
    wbytes = n_frags * 1024
    assert blob.nbytes == nfx_lib.lib.nfx_brdf_train_packed_bytes() == wbytes + 544 * 4
    fwd = ops.pack_mlp128_weights(ks, bs, nfx_lib.IN_Z_RUSINK, 1, z_dim=zd).numpy()
    assert np.array_equal(blob[:136 * 1024], fwd[:136 * 1024])          # same forward fragments as the inference blob
    rng = np.random.default_rng(6)
    K = [bf16_round(np.asarray(k, np.float32)).astype(np.float64) for k in ks]
    rows = _brdf_slot_rows(zd)
    rd = _reader(blob, wbytes, 136)
    dz = lambda f: bf16_round(rng.normal(size=(32, f)).astype(np.float32))

    def igrad(kernel_rows):
        g = dz(128)
        got = emu.dgrad_layer(rd, 8, emu.hidden_ops(g), 1)               # [32 points, 32]: C/D feature f
        for r in range(16):
            for h in range(2):
                q = 8 * (r >> 3) + (r & 3) + 4 * ((r >> 2) & 1)
                src = rows[h, q]
                want = g @ kernel_rows[src] if src >= 0 else np.zeros(32)
                _close(got[:, (r & 3) + 8 * (r >> 2) + 4 * h], want)

    g = np.zeros((32, 16), np.float32)
    g[:, :1] = dz(1)
    _close(emu.dgrad_layer(rd, 4, emu.hidden_ops(g), 4), g[:, :1] @ K[4].T)
    igrad(K[3][128:])
    for l in (3, 2, 1):
        g = dz(128)
        _close(emu.dgrad_layer(rd, 8, emu.hidden_ops(g), 4), g @ K[l][:128].T)
    igrad(K[0])
    assert rd.pos == n_frags
