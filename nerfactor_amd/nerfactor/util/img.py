"""util.img — alpha_blend and linear2srgb (reference: nerfactor/util/img.py:76-95, 140-163)."""
import numpy as np
import torch


def alpha_blend(tensor1, alpha, tensor2=None):
    """tensor1 * alpha + tensor2 * (1 - alpha); tensor2 defaults to zeros (i.e. masking)."""
    zeros_like = torch.zeros_like if isinstance(tensor1, torch.Tensor) else np.zeros_like
    if tensor2 is None:
        tensor2 = zeros_like(tensor1)
    if tensor1.ndim == 3 and alpha.ndim == 2:
        alpha = alpha[:, :, None]
    return tensor1 * alpha + tensor2 * (1. - alpha)


def linear2srgb(tensor_0to1):
    if isinstance(tensor_0to1, torch.Tensor):
        x = torch.clamp(tensor_0to1, 0., 1.)
        return torch.where(x <= 0.0031308, x * 12.92, 1.055 * torch.pow(x, 1 / 2.4) - 0.055)
    x = np.clip(tensor_0to1, 0., 1.)
    return np.where(x <= 0.0031308, x * 12.92, 1.055 * np.power(x, 1 / 2.4) - 0.055)
