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




class _SafeAtan2(torch.autograd.Function):
    """atan2(x, y) whose backward cannot go NaN at (0, 0): d/dx = y / (x^2 + y^2 + eps), d/dy = -x / (x^2 + y^2 + eps)
    (nerfactor/util/math.py:24-38)."""

    @staticmethod
    def forward(ctx, x, y, eps):
        ctx.save_for_backward(x, y)
        ctx.eps = eps
        return torch.atan2(x, y)

    @staticmethod
    def backward(ctx, dz):
        x, y = ctx.saved_tensors
        den = x * x + y * y + ctx.eps
        return dz * (y / den), dz * (-x / den), None


class _SafeAcos(torch.autograd.Function):
    """acos(clip(x, -1, 1)) with the finite slope -1 / (sqrt(1 - x^2 + eps) + eps) at +-1 (nerfactor/util/math.py:41-60)."""

    @staticmethod
    def forward(ctx, x, eps):
        xc = torch.clamp(x, -1., 1.)
        ctx.save_for_backward(xc)
        ctx.eps = eps
        return torch.acos(xc)

    @staticmethod
    def backward(ctx, dy):
        (xc,) = ctx.saved_tensors
        return dy * (-1. / (torch.sqrt(1. - xc * xc + ctx.eps) + ctx.eps)), None


def safe_atan2(x, y, eps=1e-6):
    return _SafeAtan2.apply(x, y, eps)


def safe_acos(x, eps=1e-6):
    return _SafeAcos.apply(x, eps)
