"""Thin torch-tensor front end of the C-ABI (include/nfx.h).  torch is plumbing here: it owns the
device buffers and the stream; every op below is one call into libnfx.so, enqueued on torch's
current stream.  Inputs must be CUDA (ROCm) fp32 contiguous tensors; anything else raises."""
import ctypes

import numpy as np
import torch

from . import _capi
from ._capi import PREC_BF16, PREC_FP32, check, lib

_PREC = {'bf16': PREC_BF16, 'fp32': PREC_FP32, PREC_BF16: PREC_BF16, PREC_FP32: PREC_FP32}
_ACT = {None: 0, 'none': 0, 'relu': 1, 'sigmoid': 2, 'softplus': 3}


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(t, name, shape=None):
    if t is None:
        return None
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _capi.NfxError("%s must be a CUDA/ROCm tensor (libnfx has no CPU path)" % name)
    if t.dtype != torch.float32:
        raise _capi.NfxError("%s must be float32, got %s" % (name, t.dtype))
    if not t.is_contiguous():
        t = t.contiguous()
    if shape is not None:
        if t.dim() != len(shape) or any(s is not None and s != d for s, d in zip(shape, t.shape)):
            raise _capi.NfxError("%s has shape %s, expected %s" % (name, tuple(t.shape), shape))
    return t


# ------------------------------------------------------------------------------- packing
def _as_host_f32(a):
    if isinstance(a, torch.Tensor):
        a = a.detach().cpu().numpy()
    return np.ascontiguousarray(a, dtype=np.float32)


def _pack(fn_bytes, fn_pack, kernels, biases, extra):
    ks = [_as_host_f32(k) for k in kernels]
    bs = [_as_host_f32(b) for b in biases]
    n = len(ks)
    karr = (ctypes.c_void_p * n)(*[k.ctypes.data for k in ks])
    barr = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
    nbytes = fn_bytes(*extra)
    if nbytes == 0:
        raise _capi.NfxError("packing: unsupported configuration %r" % (extra,))
    blob = np.zeros(nbytes, np.uint8)
    check(fn_pack(karr, barr, *extra, blob.ctypes.data, nbytes), fn_pack.__name__)
    return torch.from_numpy(blob)


NERF_LAYER_SHAPES = [(63, 256)] + [(256, 256)] * 4 + [(319, 256)] + [(256, 256)] * 2 + \
    [(256, 1), (256, 256), (283, 128), (128, 3)]


def pack_nerf_weights(kernels, biases, prec='bf16'):
    """kernels/biases: 12 Keras-layout arrays [in,out]/[out]: enc[0..7], sigma_out, bottleneck,
    rgb_out[0], rgb_out[1] (nerfactor/models/nerf.py:53-71).  Returns the uint8 blob (host)."""
    if len(kernels) != 12 or len(biases) != 12:
        raise _capi.NfxError("pack_nerf_weights: need 12 kernels and 12 biases")
    for i, (k, b, shp) in enumerate(zip(kernels, biases, NERF_LAYER_SHAPES)):
        if tuple(k.shape) != shp or tuple(b.shape) != (shp[1],):
            raise _capi.NfxError("pack_nerf_weights: layer %d has shapes %s/%s, expected %s/%s" % (
                i, tuple(k.shape), tuple(b.shape), shp, (shp[1],)))
    return _pack(lib.nfx_nerf_packed_bytes, lib.nfx_nerf_pack_weights, kernels, biases,
                 (_PREC[prec],))


def pack_mlp128_weights(kernels, biases, in_kind, out_dim, z_dim=0, prec='bf16'):
    """kernels/biases: the 4 mlp layers + the out layer of a width-128 surface MLP
    (nerfactor/models/shape.py:79-94, nerfactor.py:128-143, brdf.py:57-66)."""
    if len(kernels) != 5 or len(biases) != 5:
        raise _capi.NfxError("pack_mlp128_weights: need 5 kernels and 5 biases")
    return _pack(lib.nfx_mlp128_packed_bytes, lib.nfx_mlp128_pack_weights, kernels, biases,
                 (in_kind, z_dim, out_dim, _PREC[prec]))


# ------------------------------------------------------------------------------ NeRF ops
def l2_normalize3(x, eps):
    x = _dev(x, 'x', (None, 3))
    out = torch.empty_like(x)
    check(lib.nfx_l2_normalize3(_ptr(x), _ptr(out), x.shape[0], eps, _stream()), 'nfx_l2_normalize3')
    return out


def gen_z(near, far, n_samples, n_rays, lin_in_disp=False, u=None, device='cuda'):
    u = _dev(u, 'u', (n_rays, n_samples))
    z = torch.empty((n_rays, n_samples), dtype=torch.float32, device=device)
    check(lib.nfx_gen_z(near, far, n_samples, n_rays, int(lin_in_disp), _ptr(u), _ptr(z), _stream()),
          'nfx_gen_z')
    return z


def nerf_mlp_fwd(rayo, rayd, z, blob, prec='bf16'):
    """rgbs[N,S,4] = NeRF MLP at rayo + rayd*z with view direction rayd (already normalised)."""
    rayo = _dev(rayo, 'rayo', (None, 3))
    n = rayo.shape[0]
    rayd = _dev(rayd, 'rayd', (n, 3))
    z = _dev(z, 'z', (n, None))
    if not blob.is_cuda or blob.dtype != torch.uint8:
        raise _capi.NfxError("blob must be a CUDA uint8 tensor")
    s = z.shape[1]
    out = torch.empty((n, s, 4), dtype=torch.float32, device=z.device)
    check(lib.nfx_nerf_mlp_fwd(_ptr(rayo), _ptr(rayd), _ptr(z), n, s, _ptr(blob), _PREC[prec],
                               _ptr(out), _stream()), 'nfx_nerf_mlp_fwd')
    return out


def composite_fwd(rgbs, z, rayd, white_bg=True, noise=None, want_weights=True):
    rgbs = _dev(rgbs, 'rgbs', (None, None, 4))
    n, s = rgbs.shape[:2]
    z = _dev(z, 'z', (n, s))
    rayd = _dev(rayd, 'rayd', (n, 3))
    noise = _dev(noise, 'noise', (n, s))
    dev = rgbs.device
    rgb = torch.empty((n, 3), dtype=torch.float32, device=dev)
    occu = torch.empty((n,), dtype=torch.float32, device=dev)
    depth = torch.empty_like(occu)
    disp = torch.empty_like(occu)
    w = torch.empty((n, s), dtype=torch.float32, device=dev) if want_weights else None
    check(lib.nfx_composite_fwd(_ptr(rgbs), _ptr(z), _ptr(rayd), _ptr(noise), n, s, int(white_bg),
                                _ptr(rgb), _ptr(occu), _ptr(depth), _ptr(disp), _ptr(w), _stream()),
          'nfx_composite_fwd')
    return rgb, occu, depth, disp, w


def sample_fine(z, weights, n_fine, u=None):
    z = _dev(z, 'z', (None, None))
    n, nc = z.shape
    weights = _dev(weights, 'weights', (n, nc))
    u = _dev(u, 'u', (n, n_fine))
    out = torch.empty((n, nc + n_fine), dtype=torch.float32, device=z.device)
    check(lib.nfx_sample_fine(_ptr(z), _ptr(weights), n, nc, n_fine, _ptr(u), _ptr(out), _stream()),
          'nfx_sample_fine')
    return out


# --------------------------------------------------------------------------- diagnostics
def selftest_mfma_bf16(a, b):
    a = _dev(a, 'a', (32, 16))
    b = _dev(b, 'b', (16, 32))
    d = torch.empty((32, 32), dtype=torch.float32, device=a.device)
    check(lib.nfx_selftest_mfma_bf16(_ptr(a), _ptr(b), _ptr(d), _stream()), 'nfx_selftest_mfma_bf16')
    return d


def selftest_sincos(x, which):
    x = _dev(x, 'x')
    out = torch.empty_like(x)
    check(lib.nfx_selftest_sincos(_ptr(x), x.numel(), int(which), _ptr(out), _stream()),
          'nfx_selftest_sincos')
    return out
