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
    WEIGHT_BYTES = 1192 * 1024
    rd = BlobReader(np.asarray(blob), WEIGHT_BYTES)
    bias = rd.b
    pe = posenc_slots(pts.astype(np.float32), 10)
    pv = posenc_slots(views.astype(np.float32), 4)
    h = layer(rd, 4, bias, 0, pe, 8, True)
    for l in range(1, 8):
        if l == 5:
            h = layer(rd, 20, bias, 256 * l, h + pe, 8, True)
        else:
            h = layer(rd, 16, bias, 256 * l, h, 8, True)
    feat = layer(rd, 16, bias, 2048, h, 8, False)
    acc = tile(rd, 16, bias, 2048 + 256, h)
    sigma = acc[:32, 0]
    r0 = layer(rd, 20, bias, 2048 + 288, feat + pv, 4, True)
    acc = tile(rd, 8, bias, 2048 + 288 + 128, r0)
    assert rd.pos == 1192
    return np.stack([acc[:32, 0], acc[:32, 1], acc[:32, 2], sigma], -1)
