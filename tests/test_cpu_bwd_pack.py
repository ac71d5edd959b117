"""Backward-side weight blobs (training dgrad chain, density-gradient chain) checked on the CPU: the host packers'
output is pushed through the lane-level MFMA emulation of tests/emu.py and compared with plain W^T products."""
import numpy as np

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
