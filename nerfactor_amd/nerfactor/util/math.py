"""util.math — the `safe_*` helpers of the reference (nerfactor/util/math.py:63-94) on torch.
Off-hot-path conveniences; the per-ray versions live in the HIP kernels."""
import torch



def safe_l2_normalize(x, axis=None, eps=1e-6):
    """tf.linalg.l2_normalize(x, axis, epsilon=eps) = x * rsqrt(max(sum(x^2), eps))."""
    sq = torch.sum(x * x, dim=axis, keepdim=True)
    return x * torch.rsqrt(torch.clamp(sq, min=eps))


def safe_cumprod(x, eps=1e-6):
    """Exclusive cumulative product of (x + eps) along the last axis."""
    y = torch.cumprod(x + eps, -1)
    return torch.cat((torch.ones_like(y[..., :1]), y[..., :-1]), -1)


