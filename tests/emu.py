"""Lane-level NumPy emulation of the fused-MLP kernels (test infrastructure).

Re-enacts, for ONE column tile (32 points) of one wave, exactly what nerf_mlp.hip /
mlp_engine.hpp do with the packed blob: per-lane positional-encoding slots, one
v_mfma_f32_32x32x16_bf16 per (fragment, k-step) with the documented A/B/C lane maps, bias
initialisation from the permuted bias table, relu + bf16 conversion of the accumulator
registers straight into the next layer's B operand.  It lets the CPU test-suite check the host
packer and the register-dataflow design against the oracle without a GPU.
"""
import numpy as np

from oracle.nerf_ref import bf16_round

LANES = np.arange(64)
H = LANES >> 5
P = LANES & 31


def bf16_bits_to_f32(u16):
    return (u16.astype(np.uint32) << 16).view(np.float32)


def mfma_32x32x16(a_frag, b_frag, acc):
    """a_frag, b_frag [64, 8] fp32 values (already bf16-representable); acc [64, 16] fp32."""
    A = np.zeros((32, 16), np.float32)
    B = np.zeros((16, 32), np.float32)
    for j in range(8):
        A[P, 8 * H + j] = a_frag[:, j]
        B[8 * H + j, P] = b_frag[:, j]
    D = (A.astype(np.float64) @ B.astype(np.float64)).astype(np.float32)
    out = acc.copy()
    for r in range(16):
        out[:, r] += D[(r & 3) + 8 * (r >> 2) + 4 * H, P]
    return out


def posenc_slots(x, L):
    """x [32, 3] (one point per lane&31) -> list of k-step operands [64, 8] (bf16-rounded)."""
    ks = (3 * L + 2 + 7) // 8
    v = np.zeros((64, ks * 8), np.float32)
    xl = x[P]  # [64, 3]
    for q in range(ks * 8):
        if q < 3 * L:
            arg = xl[:, q % 3] * np.float32(2 ** (q // 3))
            v[:, q] = np.where(H == 1, np.cos(arg), np.sin(arg))
        elif q == 3 * L:
            v[:, q] = np.where(H == 1, xl[:, 2], xl[:, 0])
        elif q == 3 * L + 1:
            v[:, q] = np.where(H == 1, 0, xl[:, 1])
    v = bf16_round(v)
    return [v[:, 8 * s:8 * s + 8] for s in range(ks)]


class BlobReader:
    def __init__(self, blob, weight_bytes):
        self.w = np.frombuffer(blob[:weight_bytes].tobytes(), np.uint16)
        self.b = np.frombuffer(blob[weight_bytes:].tobytes(), np.float32)
        self.pos = 0  # in fragments

    def chunk(self, n_frags_padded):
        frags = self.w[self.pos * 512:(self.pos + n_frags_padded) * 512].reshape(-1, 64, 8)
        self.pos += n_frags_padded
        return bf16_bits_to_f32(frags)


def tile(reader, chunk_frags, bias, bias_off, b_ops):
    """One 32-row output tile: returns acc [64,16]."""
    frags = reader.chunk(chunk_frags)
    acc = np.zeros((64, 16), np.float32)
    for r in range(16):
        acc[:, r] = bias[bias_off + (r & 3) + 8 * (r >> 2) + 4 * H]
    for s, b in enumerate(b_ops):
        acc = mfma_32x32x16(frags[s], b, acc)
    return acc


def layer(reader, chunk_frags, bias, bias_off, b_ops, n_tiles, relu):
    out = []
    for t in range(n_tiles):
        acc = tile(reader, chunk_frags, bias, bias_off + 32 * t, b_ops)
        if relu:
            acc = np.maximum(acc, 0)
        acc = bf16_round(acc)
        out += [acc[:, :8], acc[:, 8:]]
    return out


def nerf_tile(blob, pts, views):
    """pts, views [32,3] -> raw [32,4] (rgb, sigma), emulating nerf_mlp_bf16_kernel."""
    WEIGHT_BYTES = 1272 * 1024
    rd = BlobReader(np.asarray(blob), WEIGHT_BYTES)
    bias = rd.b
    pe = posenc_slots(pts.astype(np.float32), 10)
    pv = posenc_slots(views.astype(np.float32), 4)
    h = layer(rd, 8, bias, 0, pe, 8, True)
    for l in range(1, 8):
        if l == 5:
            h = layer(rd, 24, bias, 256 * l, h + pe, 8, True)
        else:
            h = layer(rd, 16, bias, 256 * l, h, 8, True)
    feat = layer(rd, 16, bias, 2048, h, 8, False)
    acc = tile(rd, 16, bias, 2048 + 256, h)
    sigma = acc[:32, 0]
    r0 = layer(rd, 24, bias, 2048 + 288, feat + pv, 4, True)
    acc = tile(rd, 8, bias, 2048 + 288 + 128, r0)
    assert rd.pos == 1272
    return np.stack([acc[:32, 0], acc[:32, 1], acc[:32, 2], sigma], -1)


# ----------------------------------------------------------------------------- width-128 nets
M128_MAIN_W = 136 * 1024
M128_PRE_W = 32 * 1024


def _m128_mid_and_out(rd, bias, h, skip_ops, pre3=None):
    h = layer(rd, 8, bias, 128, h, 4, True)
    h = layer(rd, 8, bias, 256, h, 4, True)
    if pre3 is None:
        h = layer(rd, 12, bias, 384, h + skip_ops, 4, True)
    else:
        h = layer_pre(rd, 12, pre3, h + skip_ops, 4)
    return tile(rd, 8, bias, 512, h)


def layer_pre(reader, chunk_frags, pre_vec, b_ops, n_tiles):
    """Like layer() but the accumulators start from a per-point vector pre_vec [32 pts, 128]
    (relu, bf16 out).  All 32 lanes&31 share ONE point in the lvis kernel; here we allow a
    per-lane point for generality."""
    out = []
    for t in range(n_tiles):
        frags = reader.chunk(chunk_frags)
        acc = np.zeros((64, 16), np.float32)
        for r in range(16):
            acc[:, r] = pre_vec[P, 32 * t + (r & 3) + 8 * (r >> 2) + 4 * H]
        for s, b in enumerate(b_ops):
            acc = mfma_32x32x16(frags[s], b, acc)
        acc = bf16_round(np.maximum(acc, 0))
        out += [acc[:, :8], acc[:, 8:]]
    return out


def mlp128_xyz_tile(blob, pts, out_dim):
    """pts [32,3] (already scaled) -> raw out [32, out_dim] (pre-activation)."""
    rd = BlobReader(np.asarray(blob), M128_MAIN_W)
    pe = posenc_slots(pts.astype(np.float32), 10)
    h = layer(rd, 4, rd.b, 0, pe, 4, True)
    acc = _m128_mid_and_out(rd, rd.b, h, pe)
    assert rd.pos == 136
    rows = np.zeros((32, out_dim), np.float32)
    for row in range(out_dim):
        rows[:, row] = acc[:32, row] if row < 4 else acc[32:, row - 4]
    return rows


def lvis_tile(blob, pt, ldirs):
    """One surface point pt [3] (scaled), 32 light directions [32,3] -> raw logits [32]."""
    blob = np.asarray(blob)
    pre_blob = blob[:M128_PRE_W + 1024]
    rd = BlobReader(pre_blob, M128_PRE_W)
    pe = posenc_slots(np.broadcast_to(pt.astype(np.float32), (32, 3)), 10)
    pre = np.zeros((32, 256), np.float32)
    for t in range(8):
        acc = tile(rd, 4, rd.b, 32 * t, pe)
        for r in range(16):
            pre[P, 32 * t + (r & 3) + 8 * (r >> 2) + 4 * H] = acc[:, r]
    rd = BlobReader(blob[M128_PRE_W + 1024:], M128_MAIN_W)
    pl = posenc_slots(ldirs.astype(np.float32), 4)
    h = layer_pre(rd, 4, pre[:, :128], pl, 4)
    acc = _m128_mid_and_out(rd, rd.b, h, pl, pre3=pre[:, 128:])
    assert rd.pos == 136
    return acc[:32, 0]


def brdf_tile(blob, z, rusink):
    """z [32, zd], rusink [32, 3] -> raw logits [32] of the learned-BRDF MLP."""
    zd = z.shape[1]
    zl, rl = z[P].astype(np.float32), rusink[P].astype(np.float32)
    v = np.zeros((64, 16), np.float32)
    for q in range(6):
        arg = rl[:, q % 3] * np.float32(2 ** (q // 3))
        v[:, q] = np.where(H == 1, np.cos(arg), np.sin(arg))
    v[:, 6] = np.where(H == 1, rl[:, 2], rl[:, 0])
    v[:, 7] = np.where(H == 1, zl[:, 0], rl[:, 1])
    for j in range(8):
        for hh in range(2):
            i = 1 + 2 * j + hh
            if i < zd:
                v[H == hh, 8 + j] = zl[H == hh, i]
    v = bf16_round(v)
    ops_in = [v[:, :8], v[:, 8:]]
    rd = BlobReader(np.asarray(blob), M128_MAIN_W)
    h = layer(rd, 4, rd.b, 0, ops_in, 4, True)
    acc = _m128_mid_and_out(rd, rd.b, h, ops_in)
    assert rd.pos == 136
    return acc[:32, 0]


# ----------------------------------------------------------------------------- backward-side blobs
def hidden_ops(x):
    """Logical activations x [32, F] (F a multiple of 16) -> hidden-layout B operands: operand o holds features
    16 o + (j & 3) + 8 (j >> 2) + 4 h in element j of lane (h, p) — what relu+cvt of a C/D tile leaves in registers."""
    x = bf16_round(np.asarray(x, np.float32))
    j = np.arange(8)
    ops = []
    for o in range(x.shape[1] // 16):
        feat = 16 * o + (j & 3)[None, :] + 8 * (j >> 2)[None, :] + 4 * H[:, None]
        ops.append(x[P[:, None], feat])
    return ops


def tile_features(acc, n_valid=32):
    """C/D tile acc [64, 16] -> [32 points, 32 features] (feature = (r & 3) + 8 (r >> 2) + 4 h)."""
    out = np.zeros((32, 32), np.float32)
    for r in range(16):
        out[P, (r & 3) + 8 * (r >> 2) + 4 * H] = acc[:, r]
    return out[:, :n_valid]


def dgrad_layer(reader, chunk_frags, b_ops, n_tiles):
    """Transposed-layer product with zero-initialised accumulators: -> [32, 32 n_tiles] fp32."""
    zero = np.zeros(32 * n_tiles + 64, np.float32)
    return np.concatenate([tile_features(tile(reader, chunk_frags, zero, 0, b_ops)) for _ in range(n_tiles)], 1)


def posenc10_slot_rows():
    """Keras-kernel input row (of the 63-wide posenc(x) block) behind slot q of lane half h, or -1."""
    rows = -np.ones((2, 32), np.int64)
    for q in range(30):
        rows[0, q] = 3 + 6 * (q // 3) + (q % 3)        # sin(2^b x_c)
        rows[1, q] = 3 + 6 * (q // 3) + 3 + (q % 3)    # cos(2^b x_c)
    rows[0, 30], rows[1, 30], rows[0, 31] = 0, 2, 1
    return rows
