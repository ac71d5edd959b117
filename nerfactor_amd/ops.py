"""Thin torch-tensor front end of the C-ABI (include/nfx.h).  torch is plumbing here: it owns the
device buffers and the stream; every op below is one call into libnfx.so, enqueued on torch's
current stream.  Inputs must be CUDA (ROCm) fp32 contiguous tensors; anything else raises."""
import ctypes
import os

import numpy as np
import torch

from . import _capi
from ._capi import PREC_BF16, PREC_FP32, PREC_FP32_NATIVE, check, lib

# 'fp32' = fp32-class (bf16 hi / lo operand pairs on the bf16 matrix pipe); 'fp32_native' (runtime-shaped kernels only) = fp32
# operands on the native fp32 matrix instruction
_PREC = {'bf16': PREC_BF16, 'fp32': PREC_FP32, 'fp32_native': PREC_FP32_NATIVE, PREC_BF16: PREC_BF16, PREC_FP32: PREC_FP32,
         PREC_FP32_NATIVE: PREC_FP32_NATIVE}
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



class NotAGather(_capi.NfxError):
    """The packer's output is not a pure gather of the parameters (no device re-pack map exists for it)."""


class DevicePacker:
    """Re-packs one network's blob ON THE DEVICE (nfx_pack_gather) from an index map derived once from the host
    packer `pack_fn(kernels, biases) -> uint8 blob`: the packer is run on arrays holding the base-256 digits (+1)
    of each parameter's flat index — exactly representable in bf16 — so the blobs it returns ARE the gather map
    (0 = padding).  A probe with every parameter = 1 + 2^-20 tells fp32 words from bf16 pairs, and the LO halves of the
    fp32-class kernels' hi / lo operand pairs from the hi ones (round 5: those blobs are re-packed on the device too)."""

    def __init__(self, pack_fn, shapes_k, shapes_b):
        # a packer whose blob is NOT a gather (the hi / lo fragments of the runtime-shaped kernels' fp32-class mode) may
        # name one that is (`pack_fn.gather_fn`: the same blob with fp32 fragments) plus the device pass that turns the
        # gathered blob into its own (`pack_fn.post(blob)`, in place): the map is derived from gather_fn, the one-time
        # bit-for-bit check still runs against pack_fn itself
        self.pack_fn = pack_fn
        self.post = getattr(pack_fn, 'post', None)
        map_fn = getattr(pack_fn, 'gather_fn', pack_fn)
        self.shapes = [tuple(s) for s in list(shapes_k) + list(shapes_b)]
        self.nk = len(shapes_k)
        sizes = [int(np.prod(s)) for s in self.shapes]
        if sum(sizes) >= 1 << 24:      # (bit 30 of a map entry is the residual flag)
            raise _capi.NfxError("DevicePacker: network too large for the 3-digit index map")
        offs = np.cumsum([0] + sizes)

        def run(arrays):
            return map_fn(arrays[:self.nk], arrays[self.nk:]).numpy()

        # probe: every parameter = 1 + 2^-20 -> an fp32 word reads 0x3f800008, a bf16 half 0x3f80 (or 0 = padding); the LO
        # half of an fp32-class operand pair (mlp_x3.hpp: bf16(v - bf16(v))) reads bf16(2^-20) = 0x3580
        probe = run([np.full(s, 1. + 2. ** -20, np.float32) for s in self.shapes]).view(np.uint32)
        is_bias = probe == np.uint32(0x3f800008)                  # "bias" = any parameter stored as fp32
        kinds = (0, 0x3f80, 0x3580)
        halves_ok = np.isin(probe & np.uint32(0xffff), kinds) & np.isin(probe >> np.uint32(16), kinds)
        if not (is_bias | halves_ok).all() or not is_bias.any():
            raise NotAGather("DevicePacker: cannot separate the fp32 and bf16 regions of the blob")
        res_lo = ~is_bias & ((probe & np.uint32(0xffff)) == 0x3580)       # this half holds a residual (lo) value
        res_hi = ~is_bias & ((probe >> np.uint32(16)) == 0x3580)
        self.n_words = int(is_bias.size)
        self.nbytes = self.n_words * 4
        lo = np.zeros(self.n_words, np.int64)
        hi = np.zeros(self.n_words, np.int64)
        fb = np.zeros(self.n_words, np.int64)
        pad_lo = pad_hi = pad_b = None
        # digit passes: parameter i = (digit_d(i) + 1) (1 + 2^-10).  bf16 of it is digit + 1 exactly (the 2^-10 part is under
        # half an ulp of any 8-bit integer), its residual is (digit + 1) 2^-10 exactly, an fp32 word holds the product
        eps = np.float32(2. ** -10)
        for d in range(3):
            arrays = [((((np.arange(offs[i], offs[i + 1]) >> (8 * d)) & 255) + 1).astype(np.float32) * (np.float32(1.) + eps)).reshape(s)
                      for i, s in enumerate(self.shapes)]
            words = run(arrays).view(np.uint32)
            vlo = (words << 16).view(np.float32).astype(np.float64)              # low bf16 of every word
            vhi = (words & np.uint32(0xffff0000)).view(np.float32).astype(np.float64)
            vlo = np.rint(np.where(res_lo, vlo * 1024., vlo)).astype(np.int64)
            vhi = np.rint(np.where(res_hi, vhi * 1024., vhi)).astype(np.int64)
            vb = np.rint(np.where(is_bias, words.view(np.float32).astype(np.float64) / (1. + 2. ** -10), 0.)).astype(np.int64)
            if d == 0:
                pad_lo, pad_hi, pad_b = vlo == 0, vhi == 0, vb == 0
            lo += np.maximum(vlo - 1, 0) << (8 * d)
            hi += np.maximum(vhi - 1, 0) << (8 * d)
            fb += np.maximum(vb - 1, 0) << (8 * d)
        lo = np.where(res_lo, lo | (1 << 30), lo)
        hi = np.where(res_hi, hi | (1 << 30), hi)
        lo[pad_lo], hi[pad_hi], fb[pad_b] = -1, -1, -1
        m = np.empty((self.n_words, 2), np.int32)
        m[:, 0] = np.where(is_bias, fb, lo)
        m[:, 1] = np.where(is_bias, -2, hi)
        self.map_host = m
        self._dev = {}
        self._checked = set()

    def _map_in_place(self, tensors):
        """(base address, device map) when every parameter is a contiguous view of one storage: the map's indices into the
        concatenation, translated to element offsets from the lowest parameter's address.  None otherwise."""
        try:
            store = tensors[0].untyped_storage().data_ptr()
            if any(t.untyped_storage().data_ptr() != store or not t.is_contiguous() or t.dtype != torch.float32 for t in tensors):
                return None
        except Exception:
            return None
        ptrs = tuple(t.data_ptr() for t in tensors)
        key = (tensors[0].device, ptrs)
        hit = self._dev.get(key)
        if hit is None:
            base = min(ptrs)
            sizes = [int(np.prod(s)) for s in self.shapes]
            offs = np.cumsum([0] + sizes)
            where = np.concatenate([np.full(k, (p - base) // 4 - offs[i], np.int64) for i, (k, p) in enumerate(zip(sizes, ptrs))])
            if max((p - base) // 4 + k for k, p in zip(sizes, ptrs)) >= 1 << 30:
                return None
            m = self.map_host.astype(np.int64)
            idx = m & 0x3fffffff
            moved = np.where(m >= 0, (idx + where[np.minimum(idx, where.size - 1)]) | (m & (1 << 30)), m)
            hit = (base, torch.from_numpy(moved.astype(np.int32)).to(tensors[0].device))
            self._dev[key] = hit
        return hit

    def pack(self, tensors):
        """tensors: the network's parameters (CUDA fp32, kernels then biases) -> uint8 CUDA blob."""
        dev = tensors[0].device
        blob = torch.empty(self.nbytes, dtype=torch.uint8, device=dev)
        in_place = self._map_in_place(tensors)
        if in_place is not None:
            # the parameters are views of ONE buffer (optim.AMSGrad's flat bucket): gather straight out of it — no
            # concatenation (2 launches per blob and step, 16 of the 92 launches of a NeRFactor step)
            base, dmap = in_place
            check(lib.nfx_pack_gather(ctypes.c_void_p(base), _ptr(dmap), self.n_words, _ptr(blob), _stream()), 'nfx_pack_gather')
        else:
            if dev not in self._dev:
                self._dev[dev] = torch.from_numpy(self.map_host).to(dev)
            src = torch.cat([t.detach().reshape(-1) for t in tensors])
            check(lib.nfx_pack_gather(_ptr(src), _ptr(self._dev[dev]), self.n_words, _ptr(blob), _stream()),
                  'nfx_pack_gather')
        if self.post is not None:
            self.post(blob)
        path = 'cat' if in_place is None else ('in_place', in_place[0], tuple(t.data_ptr() for t in tensors))
        if path not in self._checked:   # once per network AND gather path (the concatenation map; every in-place map — translated
            # offsets, the bit-30 residual flag, a raw base pointer — the first time it is used): the device gather must
            # reproduce the host packer bit for bit
            want = self.pack_fn(list(tensors[:self.nk]), list(tensors[self.nk:]))
            if not torch.equal(blob.cpu(), want):
                raise _capi.NfxError("DevicePacker: device re-pack (%s path) differs from the host packer" % (
                    path if path == 'cat' else 'in-place'))
            self._checked.add(path)
        return blob


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


def l2_normalize_rows(x, eps=1e-6):
    """tf.linalg.l2_normalize(x, axis=1, epsilon=eps) of x[n, d], d <= 16 (one launch)."""
    x = _dev(x, 'x', (None, None))
    out = torch.empty_like(x)
    check(lib.nfx_l2_normalize_rows(_ptr(x), _ptr(out), x.shape[0], x.shape[1], eps, _stream()), 'nfx_l2_normalize_rows')
    return out


def l2_normalize_rows_bwd(x, dy, eps=1e-6):
    """dx of l2_normalize_rows(x) given dy = dLoss / d y (one launch)."""
    x = _dev(x, 'x', (None, None))
    dy = _dev(dy, 'dy', tuple(x.shape))
    dx = torch.empty_like(x)
    check(lib.nfx_l2_normalize_rows_bwd(_ptr(x), _ptr(dy), _ptr(dx), x.shape[0], x.shape[1], eps, _stream()),
          'nfx_l2_normalize_rows_bwd')
    return dx


def light_smoothness(light, tv_weight, achro_weight):
    """(loss[1], grad[h, w, 3]) of the light probe's smoothness penalties (nerfactor.py:526-539), one launch."""
    light = _dev(light, 'light', (None, None, 3))
    loss = torch.empty((1,), dtype=torch.float32, device=light.device)
    grad = torch.empty_like(light)
    check(lib.nfx_light_smoothness(_ptr(light), light.shape[0], light.shape[1], float(tv_weight), float(achro_weight),
                                   _ptr(loss), _ptr(grad), _stream()), 'nfx_light_smoothness')
    return loss, grad


def all_finite(x):
    """0-dim bool CUDA tensor: no Inf / NaN in the fp32 tensor `x` (one read of x; tf.debugging.check_numerics)."""
    x = _dev(x, 'x')
    flag = torch.zeros((), dtype=torch.int32, device=x.device)
    check(lib.nfx_any_nonfinite(_ptr(x), x.numel(), _ptr(flag), _stream()), 'nfx_any_nonfinite')
    return flag == 0


def scatter_rows(src, row_of, n_all=None):
    """out[i] = src[row_of[i]] where row_of[i] >= 0, zeros elsewhere (tf.scatter_nd of the foreground rows): ONE pass that
    writes every output element once.  src [m, ...] fp32, row_of [n_all] int32."""
    src = _dev(src, 'src')
    row_of = row_of.contiguous()
    if not row_of.is_cuda or row_of.dtype != torch.int32:
        raise _capi.NfxError("scatter_rows: row_of must be a CUDA int32 tensor")
    n_all = row_of.shape[0] if n_all is None else n_all
    d = 1
    for k in src.shape[1:]:
        d *= int(k)
    out = torch.empty((n_all,) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
    if n_all == 0 or d == 0:   # e.g. [n, 0, 3]: a render with no light probes
        return out
    if src.shape[0] == 0:      # no foreground row at all
        return out.zero_()
    check(lib.nfx_scatter_rows(_ptr(src), ctypes.c_void_p(row_of.data_ptr()), n_all, d, _ptr(out), _stream()),
          'nfx_scatter_rows')
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


# ------------------------------------------------------------------------------ runtime-shaped MLP (mlp_generic.hip)
class GenericNet:
    """A packed mlp.Network of arbitrary shape for mlp_generic_fwd / mlp_generic_bwd: blob (uint8 tensor, move it with
    .to(device)) + the layer description the C-ABI takes.  skip_at: the reference's list (after layer i the input is
    re-concatenated, y first: nerfactor/networks/mlp.py:47-48).  train = True packs the train blob (forward fragments +
    transposed fragments for the backward); the forward accepts either.  prec = 'fp32': fp32-class — fp32 activations and
    gradients, every MFMA operand a bf16 hi / lo pair (fragments pre-split by the packer); prec = 'fp32_native': fp32
    fragments for the native fp32 matrix instruction (nothing rounded to bf16, ~5x the matrix time)."""

    def __init__(self, kernels, biases, acts, skip_at=None, train=False, prec='bf16'):
        ks = [_as_host_f32(k) for k in kernels]
        bs = [_as_host_f32(b) for b in biases]
        n = len(ks)
        if n == 0 or len(bs) != n or len(acts) != n:
            raise _capi.NfxError("GenericNet: kernels, biases and acts must have the same, non-zero length")
        self.widths = [int(k.shape[1]) for k in ks]
        skip_at = set(skip_at or [])
        if any(i < 0 or i >= n - 1 for i in skip_at):
            # (a skip behind the last layer widens the network's OUTPUT to concat(y, x): the kernel has no such mode, and
            # dropping it silently would evaluate another function — the caller appends x itself, models/nerf.py:_enc_out)
            raise _capi.NfxError("GenericNet: skip_at %s — a skip must sit behind one of the layers 0 .. %d" % (sorted(skip_at), n - 2))
        self.skip_input = [1 if (i - 1) in skip_at else 0 for i in range(n)]
        self.d_in = int(ks[0].shape[0])
        for i in range(n):
            want = (self.d_in if i == 0 else self.widths[i - 1] + (self.d_in if self.skip_input[i] else 0))
            if ks[i].shape[0] != want or bs[i].shape != (self.widths[i],):
                raise _capi.NfxError("GenericNet: layer %d has kernel %s / bias %s, expected [%d, %d] / [%d]" % (
                    i, ks[i].shape, bs[i].shape, want, self.widths[i], self.widths[i]))
        self.acts = [_ACT[a] for a in acts]
        self._w = (ctypes.c_int * n)(*self.widths)
        self._s = (ctypes.c_int * n)(*self.skip_input)
        self._a = (ctypes.c_int * n)(*self.acts)
        size_fn, pack_fn = ((lib.nfx_mlp_generic_train_packed_bytes, lib.nfx_mlp_generic_pack_train) if train else
                            (lib.nfx_mlp_generic_packed_bytes, lib.nfx_mlp_generic_pack))
        self.prec = _PREC[prec]
        nbytes = size_fn(self.d_in, n, self._w, self._s, self.prec)
        if nbytes == 0:
            raise _capi.NfxError("GenericNet: shape outside the generic kernel's limits: " + _capi.last_error())
        blob = np.zeros(nbytes, np.uint8)
        karr = (ctypes.c_void_p * n)(*[k.ctypes.data for k in ks])
        barr = (ctypes.c_void_p * n)(*[b.ctypes.data for b in bs])
        check(pack_fn(karr, barr, self.d_in, n, self._w, self._s, self.prec, blob.ctypes.data, nbytes), 'nfx_mlp_generic_pack')
        self.blob = torch.from_numpy(blob)
        self.n_layers, self.d_out, self.train = n, self.widths[-1], bool(train)
        self.in_dims = [self.d_in if i == 0 else self.widths[i - 1] + (self.d_in if self.skip_input[i] else 0)
                        for i in range(n)]

    def to(self, device):
        self.blob = self.blob.to(device)
        return self


def generic_split_hilo(blob, net):
    """In place, on the device: the fragments of a prec = 'fp32_native' blob of `net`'s shape -> prec = 'fp32' (hi / lo pairs)."""
    if not blob.is_cuda or blob.dtype != torch.uint8:
        raise _capi.NfxError("generic_split_hilo: blob must be a CUDA uint8 tensor")
    check(lib.nfx_mlp_generic_split_hilo(_ptr(blob), net.d_in, net.n_layers, net._w, net._s, int(net.train), _stream()),
          'nfx_mlp_generic_split_hilo')
    return blob


def generic_pack_fn(acts, skip_at, train, prec, descs, tag):
    """The `pack_fn(kernels, biases) -> host blob` the models hand to their blob cache (models/base.py:_packed) for a
    runtime-shaped network; the first call leaves the GenericNet (layer description) in descs[tag].  fp32-class blobs
    carry what ops.DevicePacker needs to rebuild them on the device: the native-fp32 packer (a pure gather) and the
    in-place hi / lo split."""
    def pack(k, b):
        g = GenericNet(k, b, acts, skip_at, train=train, prec=prec)
        descs.setdefault(tag, g)
        return g.blob
    if _PREC[prec] == PREC_FP32:
        shape = {}       # (the layer description nfx_mlp_generic_split_hilo takes: the same for both fp32 layouts)

        def gather(k, b):
            g = GenericNet(k, b, acts, skip_at, train=train, prec='fp32_native')
            shape.setdefault('net', g)
            return g.blob
        pack.gather_fn = gather
        pack.post = lambda blob: generic_split_hilo(blob, shape['net'])
    return pack


def _check_out(who, out, n, col0, width, device):
    """An `out` / `col0` destination: CUDA fp32 [n, >= col0 + width] with unit column stride on `device`."""
    if not isinstance(out, torch.Tensor) or not out.is_cuda or out.dtype != torch.float32 or out.dim() != 2 \
            or out.shape[0] != n or out.stride(1) != 1 or out.device != device:
        raise _capi.NfxError("%s: out must be a CUDA fp32 [%d, ld] matrix with unit column stride on %s" % (who, n, device))
    if col0 < 0 or col0 + width > out.shape[1]:
        raise _capi.NfxError("%s: columns [%d, %d) do not fit out's %d columns" % (who, col0, col0 + width, out.shape[1]))


def mlp_generic_fwd(x, net, out=None, col0=0):
    """y[n, d_out] = net(x[n, >= d_in]) through the runtime-shaped fused kernel; `out` / `col0`: write into columns
    [col0, col0 + d_out) of an existing [n, ld] matrix (assembling a concatenation without a copy)."""
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise _capi.NfxError("mlp_generic_fwd: x must be a CUDA fp32 matrix with unit column stride")
    n = x.shape[0]
    if x.shape[1] < net.d_in:
        raise _capi.NfxError("mlp_generic_fwd: x has %d columns, the network reads %d" % (x.shape[1], net.d_in))
    if out is None:
        out = torch.empty((n, col0 + net.d_out), dtype=torch.float32, device=x.device)
    _check_out('mlp_generic_fwd', out, n, col0, net.d_out, x.device)
    check(lib.nfx_mlp_generic_fwd(_ptr(x), n, x.stride(0) if n else net.d_in, net.d_in, net.n_layers, net._w, net._a,
                                  net._s, _ptr(net.blob), net.prec, _ptr(out), out.stride(0) if n else out.shape[1],
                                  col0, _stream()), 'nfx_mlp_generic_fwd')
    return out


def mlp_generic_bwd(x, net, dy, dkernels, dbiases, want_dx=False):
    """Backward of mlp_generic_fwd(x, net): ADDS dLoss/dW into dkernels[i] ([in_i, widths[i]]) and dLoss/db into
    dbiases[i] given dy[n, d_out] = dLoss/d y; returns dLoss/dx [n, d_in] when want_dx (chained networks), else None.
    `net` must carry a train blob (GenericNet(..., train=True))."""
    if not net.train:
        raise _capi.NfxError("mlp_generic_bwd: the network was packed without the backward fragments (train=True)")
    if not isinstance(x, torch.Tensor) or not x.is_cuda or x.dtype != torch.float32 or x.dim() != 2 or x.stride(1) != 1:
        raise _capi.NfxError("mlp_generic_bwd: x must be a CUDA fp32 matrix with unit column stride")
    n = x.shape[0]
    if x.shape[1] < net.d_in:
        raise _capi.NfxError("mlp_generic_bwd: x has %d columns, the network reads %d" % (x.shape[1], net.d_in))
    dy = _dev(dy, 'dy', (n, net.d_out))
    if dkernels is None and dbiases is None:          # input gradient of a frozen network
        if not want_dx:
            raise _capi.NfxError("mlp_generic_bwd: nothing to compute (no gradient buffers, want_dx = False)")
        dkernels = dbiases = ()
    elif len(dkernels) != net.n_layers or len(dbiases) != net.n_layers:
        raise _capi.NfxError("mlp_generic_bwd: need %d kernel and bias gradient buffers" % net.n_layers)
    for i in range(len(dkernels)):
        for t, name, shape in ((dkernels[i], 'dkernels', (net.in_dims[i], net.widths[i])), (dbiases[i], 'dbiases', (net.widths[i],))):
            if _dev(t, '%s[%d]' % (name, i), shape) is not t:
                raise _capi.NfxError("mlp_generic_bwd: %s[%d] must be contiguous (it is accumulated into)" % (name, i))
    dx = torch.empty((n, net.d_in), dtype=torch.float32, device=x.device) if want_dx else None
    nbytes = lib.nfx_mlp_generic_bwd_workspace_bytes(n, net.d_in, net.n_layers, net._w, net._s, net.prec)
    ws = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=x.device)
    karr = (ctypes.c_void_p * net.n_layers)(*[t.data_ptr() for t in dkernels]) if dkernels else None
    barr = (ctypes.c_void_p * net.n_layers)(*[t.data_ptr() for t in dbiases]) if dbiases else None
    check(lib.nfx_mlp_generic_bwd(_ptr(x), n, x.stride(0) if n else net.d_in, net.d_in, net.n_layers, net._w, net._a,
                                  net._s, _ptr(net.blob), net.prec, _ptr(dy), net.d_out, 0, _ptr(dx), net.d_in, karr,
                                  barr, _ptr(ws), ws.numel(), _stream()), 'nfx_mlp_generic_bwd')
    return dx


def embed_bwd(n_freqs, x, d_out, incl_input=True, col0=0):
    """dx[n, 3] = (d embed(x) / d x)^T d_out[:, col0 : col0 + 3 incl_input + 6 n_freqs] for explicit vectors x[n, 3]."""
    x = _dev(x, 'x', (None, 3))
    if not isinstance(d_out, torch.Tensor) or not d_out.is_cuda or d_out.dtype != torch.float32 or d_out.dim() != 2 \
            or d_out.stride(1) != 1 or d_out.shape[0] != x.shape[0]:
        raise _capi.NfxError("embed_bwd: d_out must be a CUDA fp32 matrix with one row per vector")
    dx = torch.empty_like(x)
    check(lib.nfx_embed_bwd(_ptr(x), x.shape[0], n_freqs, int(incl_input), _ptr(d_out), d_out.stride(0) if x.shape[0] else 0,
                            col0, _ptr(dx), _stream()), 'nfx_embed_bwd')
    return dx


def embed(n_freqs, incl_input=True, x=None, rayo=None, rayd=None, z=None, per_ray=1, out=None, col0=0, lights=None):
    """Embedder (embedder.py:23-47) on the device: of x[n, 3] (per_ray rows per vector), of the points rayo + rayd z
    (z[n_rays, S] -> n_rays S rows), of the ray directions (rayd with per_ray rows each), or — lights[L, 3] given — of the
    unit directions from every point x[n] to every light (n L rows, shape.py:128-131).  Returns / fills
    out[:, col0 : col0 + 3 incl_input + 6 n_freqs]."""
    d_out = (3 if incl_input else 0) + 6 * n_freqs
    if lights is not None:
        mode, a, b, c = 3, _dev(x, 'x', (None, 3)), _dev(lights, 'lights', (None, 3)), None
        per_ray = b.shape[0]
        n = a.shape[0] * per_ray
    elif z is not None:
        mode, n, per_ray = 1, z.numel(), z.shape[1]
        a, b, c = _dev(rayo, 'rayo', (None, 3)), _dev(rayd, 'rayd', (None, 3)), _dev(z, 'z')
    elif x is not None:
        mode, a, b, c = 0, _dev(x, 'x', (None, 3)), None, None
        n = a.shape[0] * per_ray
    else:
        mode, a, b, c = 2, None, _dev(rayd, 'rayd', (None, 3)), None
        n = b.shape[0] * per_ray
    dev_ = (a if a is not None else b).device
    if out is None:
        out = torch.empty((n, col0 + d_out), dtype=torch.float32, device=dev_)
    _check_out('embed', out, n, col0, d_out, dev_)
    check(lib.nfx_embed(_ptr(a), _ptr(b), _ptr(c), n, per_ray, mode, n_freqs, 1 if incl_input else 0, _ptr(out),
                        out.stride(0) if n else d_out, col0, _stream()), 'nfx_embed')
    return out


def selftest_tr16(h=None, z=None):
    """h, z: [16, 32] row-major tiles -> D[32, 32] = h^T z through LDS + ds_read_b64_tr_b16 (csrc/tr16.hpp); no arguments:
    the instruction's raw lane map, [64, 4]."""
    if h is None:
        d = torch.empty((64, 4), dtype=torch.float32, device='cuda')
        check(lib.nfx_selftest_tr16(None, None, _ptr(d), 1, _stream()), 'nfx_selftest_tr16')
        return d
    h, z = _dev(h, 'h', (16, 32)), _dev(z, 'z', (16, 32))
    d = torch.empty((32, 32), dtype=torch.float32, device=h.device)
    check(lib.nfx_selftest_tr16(_ptr(h), _ptr(z), _ptr(d), 0, _stream()), 'nfx_selftest_tr16')
    return d


def selftest_sincos(x, which):
    x = _dev(x, 'x')
    out = torch.empty_like(x)
    check(lib.nfx_selftest_sincos(_ptr(x), x.numel(), int(which), _ptr(out), _stream()),
          'nfx_selftest_sincos')
    return out


# ------------------------------------------------------------------------- NeRFactor ops
def mlp128_xyz_fwd(xyz, blob, out_dim, out_act=None, xyz_scale=1., post_scale=1., post_bias=0.,
                   prec='bf16'):
    """out[n, out_dim] = post_scale * act(out(mlp(posenc10(xyz_scale * xyz)))) + post_bias."""
    xyz = _dev(xyz, 'xyz', (None, 3))
    n = xyz.shape[0]
    out = torch.empty((n, out_dim), dtype=torch.float32, device=xyz.device)
    check(lib.nfx_mlp128_xyz_fwd(_ptr(xyz), n, xyz_scale, _ptr(blob), out_dim, _ACT[out_act],
                                 post_scale, post_bias, _PREC[prec], _ptr(out), _stream()),
          'nfx_mlp128_xyz_fwd')
    return out


def lvis_rows_supported(prec='bf16'):
    """Does a render store its visibilities (and OLAT renders) straight into the rows of the full-size buffers
    (nfx_lvis_fwd_rows / nfx_shade_olat_fwd_rows)?  OPT-IN: nfx_set_option("lvis_rows", 1) — measured on one box, 20 steps
    each, A/B/A/B (profiles/r06/render_rows_ab.txt): NeRFactor-microfacet render 21.48 / 21.31 ms with, 22.10 / 21.15 without;
    learned BRDF 31.26 / 30.94 against 30.72 / 30.48; OLAT view 22.57 / 22.41 against 22.79 / 22.81 — the scatter and
    check_numerics passes it removes (~0.8 ms of traffic) are paid back by a 1 % slower kernel and the zero-fill: no gain worth
    a second two-waves-per-SIMD kernel form as the default.  bf16 kernels with the network resident in LDS only."""
    return prec == 'bf16' and _capi.get_option("lvis_variant") in (None, 8, 2, 3, 4) and _capi.get_option("lvis_rows") == 1


def zero_rows(dst, row_of):
    """dst[i, :] = 0 where row_of[i] < 0 (the background rows of a tf.scatter_nd whose foreground rows a kernel writes itself)."""
    dst = _dev(dst, 'dst', (row_of.shape[0], None))
    if not row_of.is_cuda or row_of.dtype != torch.int32 or not row_of.is_contiguous():
        raise _capi.NfxError("zero_rows: row_of must be a contiguous CUDA int32 tensor")
    check(lib.nfx_zero_rows(_ptr(dst), ctypes.c_void_p(row_of.data_ptr()), dst.shape[0], dst.shape[1], _stream()), 'nfx_zero_rows')
    return dst


def lvis_fwd(xyz, lxyz, blob, xyz_scale=1., xyz_dir=None, prec='bf16', out=None, out_row=None, nan_flag=None):
    """lvis[n, L]: the light-visibility MLP for every (surface point, light) pair.  `xyz_dir`
    (default xyz) are the points the light directions are taken from.
    Round 6: with `out` [n_all, L] and `out_row` [n] int32 the visibilities of point i are stored at out[out_row[i]] — the
    zero-filled scatter of shape.py:171-176 without its pass (the caller zeroes the other rows: zero_rows) — and `nan_flag`
    (int32[1], zeroed by the caller) receives a 1 if any of them is NaN (tf.debugging.check_numerics without its pass)."""
    xyz = _dev(xyz, 'xyz', (None, 3))
    xyz_dir = _dev(xyz_dir, 'xyz_dir', (xyz.shape[0], 3))
    lxyz = _dev(lxyz, 'lxyz', (None, 3))
    n, nl = xyz.shape[0], lxyz.shape[0]
    ws_bytes = 0 if prec == 'fp32' else lib.nfx_lvis_workspace_bytes(n)   # (the fp32-class kernel has no per-point fold)
    ws = torch.empty((max(ws_bytes, 16) // 4,), dtype=torch.float32, device=xyz.device)
    if out_row is not None:
        if out is None or not out.is_cuda or out.dtype != torch.float32 or not out.is_contiguous() or out.dim() != 2 or out.shape[1] != nl:
            raise _capi.NfxError("lvis_fwd: `out` must be a contiguous CUDA float32 [n_all, %d] tensor" % nl)
        if not out_row.is_cuda or out_row.dtype != torch.int32 or not out_row.is_contiguous() or out_row.shape != (n,):
            raise _capi.NfxError("lvis_fwd: `out_row` must be a contiguous CUDA int32 [%d] tensor" % n)
        check(lib.nfx_lvis_fwd_rows(_ptr(xyz), _ptr(xyz_dir), n, xyz_scale, _ptr(lxyz), nl, _ptr(blob), _PREC[prec], _ptr(ws),
                                    ws.numel() * 4, ctypes.c_void_p(out_row.data_ptr()), _ptr(out),
                                    None if nan_flag is None else ctypes.c_void_p(nan_flag.data_ptr()), _stream()), 'nfx_lvis_fwd_rows')
        return out
    out = torch.empty((n, nl), dtype=torch.float32, device=xyz.device)
    check(lib.nfx_lvis_fwd(_ptr(xyz), _ptr(xyz_dir), n, xyz_scale, _ptr(lxyz), nl, _ptr(blob), _PREC[prec], _ptr(ws),
                           ws.numel() * 4, _ptr(out), _stream()), 'nfx_lvis_fwd')
    _lvis_verify(out, xyz, xyz_dir, xyz_scale, lxyz, blob, prec)
    return out


_lvis_launches = [0]


def _lvis_verify(out, xyz, xyz_dir, xyz_scale, lxyz, blob, prec, every=100, max_points=8192):
    """nfx_set_option("lvis_verify", k): every k-th launch of the DEFAULT light-visibility kernel (resident128_kernel<2, 0, 8>:
    two waves per SIMD — the regime in which a sibling kernel, brdf_compact_kernel<2, 0, 8>, returns wrong rows for a reason
    three rounds of probes have not found) is checked against the one-wave-per-SIMD kernel (variant 4)
    on every `every`-th point: the two are the same sums in the same order, so ANY differing bit raises NfxError.  Costs
    ~1 % of the launch plus one host read; skipped while a hipGraph is being captured (nothing may wait there)."""
    k = _capi.get_option("lvis_verify")
    if not k or k <= 0 or prec != 'bf16' or out.shape[0] == 0:
        return
    variant = _capi.get_option("lvis_variant")
    if variant not in (None, 8) or torch.cuda.is_current_stream_capturing():
        return
    _lvis_launches[0] += 1
    if _lvis_launches[0] % k:
        return
    n = out.shape[0]
    step = max(1, min(every, n), -(-n // max_points))
    pick = torch.arange(0, n, step, device=out.device)
    x, xd = xyz[pick].contiguous(), (xyz if xyz_dir is None else xyz_dir)[pick].contiguous()
    nl = lxyz.shape[0]
    ref = torch.empty((x.shape[0], nl), dtype=torch.float32, device=out.device)
    ws = torch.empty((max(lib.nfx_lvis_workspace_bytes(x.shape[0]), 16) // 4,), dtype=torch.float32, device=out.device)
    with _capi.option("lvis_variant", 4):
        check(lib.nfx_lvis_fwd(_ptr(x), _ptr(xd), x.shape[0], xyz_scale, _ptr(lxyz), nl, _ptr(blob), _PREC[prec], _ptr(ws),
                               ws.numel() * 4, _ptr(ref), _stream()), 'nfx_lvis_fwd(verify)')
    got = out[pick]
    if not torch.equal(got, ref):
        bad = (got != ref)
        rows = torch.nonzero(bad.any(1))[:, 0]
        raise _capi.NfxError("lvis_verify: resident128_kernel<2, 0, 8> differs from the one-wave-per-SIMD kernel on %d of %d checked "
                             "points (%d elements, first point %d, max |diff| %.3e) — set lvis_variant = 4" % (
                                 rows.numel(), x.shape[0], int(bad.sum()), int(pick[rows[0]]), float((got - ref).abs().max())))


def brdf_spec_fwd(xyz, cam, normal, z, lxyz, blob, prec='bf16'):
    """spec[n, L]: learned-BRDF specular term (0 for back-lit directions)."""
    xyz = _dev(xyz, 'xyz', (None, 3))
    n = xyz.shape[0]
    cam = _dev(cam, 'cam', (n, 3))
    normal = _dev(normal, 'normal', (n, 3))
    z = _dev(z, 'z', (n, None))
    lxyz = _dev(lxyz, 'lxyz', (None, 3))
    nl = lxyz.shape[0]
    out = torch.empty((n, nl), dtype=torch.float32, device=xyz.device)
    check(lib.nfx_brdf_spec_fwd(_ptr(xyz), _ptr(cam), _ptr(normal), _ptr(z), z.shape[1], _ptr(lxyz), nl,
                                _ptr(blob), _PREC[prec], n, _ptr(out), _stream()), 'nfx_brdf_spec_fwd')
    return out


def _shade_common(xyz, cam, normal, albedo, rough, spec, lvis, lxyz, lareas, lvis_row=None):
    xyz = _dev(xyz, 'xyz', (None, 3))
    n = xyz.shape[0]
    lxyz = _dev(lxyz, 'lxyz', (None, 3))
    nl = lxyz.shape[0]
    if rough is not None and rough.dim() == 2:
        rough = rough.reshape(-1)
    if lvis_row is not None and (not lvis_row.is_cuda or lvis_row.dtype != torch.int32 or not lvis_row.is_contiguous()
                                 or lvis_row.shape != (n,)):
        raise _capi.NfxError("shade: `lvis_row` must be a contiguous CUDA int32 [%d] tensor" % n)
    return (xyz, _dev(cam, 'cam', (n, 3)), _dev(normal, 'normal', (n, 3)),
            _dev(albedo, 'albedo', (n, 3)), _dev(rough, 'rough', (n,)), _dev(spec, 'spec', (n, nl)),
            _dev(lvis, 'lvis', (n if lvis_row is None else None, nl)), lxyz, _dev(lareas.reshape(-1), 'lareas', (nl,)), n, nl)


def shade_fwd(xyz, cam, normal, albedo, lvis, lxyz, lareas, lights, rough=None, spec=None,
              spec_scale=1., f0=0.04, linear2srgb=True, lvis_row=None):
    """rgb[n, P, 3] for P lights [P, L, 3] (microfacet BRDF if `rough`, else albedo/pi + spec).  `lvis_row` [n] int32: the
    visibilities of point i are row lvis_row[i] of `lvis` (a full-size buffer written by lvis_fwd(out=, out_row=))."""
    xyz, cam, normal, albedo, rough, spec, lvis, lxyz, lareas, n, nl = _shade_common(
        xyz, cam, normal, albedo, rough, spec, lvis, lxyz, lareas, lvis_row)
    row_p = None if lvis_row is None else ctypes.c_void_p(lvis_row.data_ptr())
    lights = _dev(lights, 'lights', (None, nl, 3))
    p_total = lights.shape[0]
    out = torch.empty((n, p_total, 3), dtype=torch.float32, device=xyz.device)
    # all probes of one call must fit the LDS together; split if they do not
    p_max = p_total
    while p_max > 1 and lib.nfx_shade_lds_bytes(nl, p_max) > 160 * 1024:
        p_max -= 1
    for p0 in range(0, p_total, p_max):
        chunk = lights[p0:p0 + p_max].contiguous()
        dst = out if p_max == p_total else torch.empty((n, chunk.shape[0], 3), dtype=torch.float32,
                                                      device=xyz.device)
        check(lib.nfx_shade_fwd_rows(_ptr(xyz), _ptr(cam), _ptr(normal), _ptr(albedo), _ptr(rough), _ptr(spec),
                                     spec_scale, f0, _ptr(lvis), row_p, _ptr(lxyz), _ptr(lareas), _ptr(chunk), n, nl,
                                     chunk.shape[0], int(linear2srgb), _ptr(dst), _stream()), 'nfx_shade_fwd')
        if dst is not out:
            out[:, p0:p0 + chunk.shape[0]] = dst
    return out


def shade_olat_fwd(xyz, cam, normal, albedo, lvis, lxyz, lareas, olat_inten, ambient, rough=None,
                   spec=None, spec_scale=1., f0=0.04, linear2srgb=True, lvis_row=None, out=None, out_row=None, nan_flag=None):
    """rgb_olat[n, L, 3]: one-light-at-a-time relighting (`lvis_row`: as in shade_fwd).  `out` [n_all, L, 3] + `out_row` [n]
    int32: the renders of point i go to out[out_row[i]] (the caller zeroes the other rows); `nan_flag` int32[1]: raised when a
    radiance is NaN before the clip."""
    xyz, cam, normal, albedo, rough, spec, lvis, lxyz, lareas, n, nl = _shade_common(
        xyz, cam, normal, albedo, rough, spec, lvis, lxyz, lareas, lvis_row)
    if out_row is not None:
        if out is None or not out.is_cuda or out.dtype != torch.float32 or not out.is_contiguous() or tuple(out.shape[1:]) != (nl, 3):
            raise _capi.NfxError("shade_olat_fwd: `out` must be a contiguous CUDA float32 [n_all, %d, 3] tensor" % nl)
        if not out_row.is_cuda or out_row.dtype != torch.int32 or not out_row.is_contiguous() or out_row.shape != (n,):
            raise _capi.NfxError("shade_olat_fwd: `out_row` must be a contiguous CUDA int32 [%d] tensor" % n)
    else:
        out = torch.empty((n, nl, 3), dtype=torch.float32, device=xyz.device)
    as_p = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
    check(lib.nfx_shade_olat_fwd_rows(_ptr(xyz), _ptr(cam), _ptr(normal), _ptr(albedo), _ptr(rough), _ptr(spec),
                                      spec_scale, f0, _ptr(lvis), as_p(lvis_row), _ptr(lxyz), _ptr(lareas), olat_inten, ambient,
                                      n, nl, int(linear2srgb), as_p(out_row), _ptr(out), as_p(nan_flag), _stream()),
          'nfx_shade_olat_fwd')
    return out


def brdf_rows_geom_fwd(xyz, cam, normal, z, lxyz, n_freqs):
    """(rows [n L, z_dim + 3 + 6 n_freqs], front [n L]) of the learned BRDF in fp32: rows = [z | embed(rusink)] per (point,
    light), front = 1.0 where the light is in front of the surface (nerfactor.py:413-436)."""
    xyz = _dev(xyz, 'xyz', (None, 3))
    n = xyz.shape[0]
    cam, normal = _dev(cam, 'cam', (n, 3)), _dev(normal, 'normal', (n, 3))
    z = _dev(z, 'z', (n, None))
    lxyz = _dev(lxyz, 'lxyz', (None, 3))
    nl, zd = lxyz.shape[0], z.shape[1]
    rows = torch.empty((n * nl, zd + 3 + 6 * n_freqs), dtype=torch.float32, device=xyz.device)
    front = torch.empty((n * nl,), dtype=torch.float32, device=xyz.device)
    check(lib.nfx_brdf_rows_geom_fwd(_ptr(xyz), _ptr(cam), _ptr(normal), _ptr(z), zd, _ptr(lxyz), nl, n, n_freqs, _ptr(rows),
                                     rows.shape[1], _ptr(front), _stream()), 'nfx_brdf_rows_geom_fwd')
    return rows, front


def brdf_rows_geom_bwd(xyz, cam, normal, z_dim, lxyz, n_freqs, d_rows):
    """(d_normal [n, 3], d_z [n, z_dim]) from dLoss/d rows of brdf_rows_geom_fwd, summed over each point's front-lit lights."""
    xyz = _dev(xyz, 'xyz', (None, 3))
    n = xyz.shape[0]
    cam, normal = _dev(cam, 'cam', (n, 3)), _dev(normal, 'normal', (n, 3))
    lxyz = _dev(lxyz, 'lxyz', (None, 3))
    nl = lxyz.shape[0]
    d_rows = _dev(d_rows, 'd_rows', (n * nl, z_dim + 3 + 6 * n_freqs))
    d_normal = torch.empty((n, 3), dtype=torch.float32, device=xyz.device)
    d_z = torch.empty((n, z_dim), dtype=torch.float32, device=xyz.device)
    check(lib.nfx_brdf_rows_geom_bwd(_ptr(xyz), _ptr(cam), _ptr(normal), z_dim, _ptr(lxyz), nl, n, n_freqs, _ptr(d_rows),
                                     d_rows.shape[1], _ptr(d_normal), _ptr(d_z), _stream()), 'nfx_brdf_rows_geom_bwd')
    return d_normal, d_z


def dir2rusink(a, b):
    a = _dev(a, 'a', (None, 3))
    b = _dev(b, 'b', (a.shape[0], 3))
    out = torch.empty_like(a)
    check(lib.nfx_dir2rusink(_ptr(a), _ptr(b), a.shape[0], _ptr(out), _stream()), 'nfx_dir2rusink')
    return out


# ---------------------------------------------------------------------------- training ops
def pack_mlp128_train_weights(kernels, biases, in_kind, out_dim, prec='bf16'):
    """Train blob (forward + dgrad fragments) of a width-128 surface MLP, for mlp128_bwd."""
    if len(kernels) != 5 or len(biases) != 5:
        raise _capi.NfxError("pack_mlp128_train_weights: need 5 kernels and 5 biases")
    return _pack(lambda k, o, p: lib.nfx_mlp128_train_packed_bytes(k), lib.nfx_mlp128_pack_train_weights,
                 kernels, biases, (in_kind, out_dim, _PREC[prec]))


def mlp128_bwd(in_kind, xyz, dout, blob, dkernels, dbiases, out_act=None, xyz_scale=1., post_scale=1.,
               lxyz=None, xyz_dir=None, prec='bf16'):
    """Accumulate the weight gradients of one width-128 MLP call into `dkernels` / `dbiases`
    (lists of 5 fp32 CUDA tensors, Keras layout) given dout = dLoss/d(output)."""
    xyz = _dev(xyz, 'xyz', (None, 3))
    n = xyz.shape[0]
    nl = 0
    if in_kind == _capi.IN_XYZ_LDIR:
        lxyz = _dev(lxyz, 'lxyz', (None, 3))
        nl = lxyz.shape[0]
    rows = n if in_kind == _capi.IN_XYZ else n * nl
    dout = _dev(dout.reshape(rows, -1), 'dout', (rows, None))
    out_dim = dout.shape[1]
    xyz_dir = _dev(xyz_dir, 'xyz_dir', (n, 3))
    for t in list(dkernels) + list(dbiases):
        _dev(t, 'gradient buffer')
    ws_bytes = lib.nfx_mlp128_bwd_workspace_bytes(in_kind, n, nl)
    ws = torch.empty((max(ws_bytes, 16) // 2,), dtype=torch.bfloat16, device=xyz.device)
    karr = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in dkernels])
    barr = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in dbiases])
    check(lib.nfx_mlp128_bwd(in_kind, _ptr(xyz), _ptr(xyz_dir), n, xyz_scale, _ptr(lxyz), nl, _ptr(blob),
                             out_dim, _ACT[out_act], post_scale, _ptr(dout), _ptr(ws), ws.numel() * 2, karr,
                             barr, _PREC[prec], _stream()), 'nfx_mlp128_bwd')
    return ws



MLP128_MAX_HEADS = 4


def mlp128_bwd_heads(in_kind, xyz, heads, xyz_scale=1., lxyz=None, xyz_dir=None, prec='bf16'):
    """mlp128_bwd for up to MLP128_MAX_HEADS networks over the SAME rows in one launch pair (nfx_mlp128_bwd_heads).
    heads: [(dout, train_blob, dkernels, dbiases, out_act, post_scale), ...]; bit-identical to one mlp128_bwd per head."""
    if not 1 <= len(heads) <= MLP128_MAX_HEADS:
        raise _capi.NfxError("mlp128_bwd_heads: 1 .. %d heads" % MLP128_MAX_HEADS)
    xyz = _dev(xyz, 'xyz', (None, 3))
    n = xyz.shape[0]
    nl = 0
    if in_kind == _capi.IN_XYZ_LDIR:
        lxyz = _dev(lxyz, 'lxyz', (None, 3))
        nl = lxyz.shape[0]
    rows = n if in_kind == _capi.IN_XYZ else n * nl
    xyz_dir = _dev(xyz_dir, 'xyz_dir', (n, 3))
    douts, keep = [], []
    for dout, blob, dks, dbs, _, _ in heads:
        douts.append(_dev(dout.reshape(rows, -1), 'dout', (rows, None)))
        for t in list(dks) + list(dbs):
            _dev(t, 'gradient buffer')
        if len(dks) != 5 or len(dbs) != 5:
            raise _capi.NfxError("mlp128_bwd_heads: 5 kernel and 5 bias gradient buffers per head")
    nh = len(heads)
    ws_bytes = lib.nfx_mlp128_bwd_workspace_bytes(in_kind, n, nl) * nh
    ws = torch.empty((max(ws_bytes, 16) // 2,), dtype=torch.bfloat16, device=xyz.device)
    blobs = (ctypes.c_void_p * nh)(*[h[1].data_ptr() for h in heads])
    darr = (ctypes.c_void_p * nh)(*[d.data_ptr() for d in douts])
    dims = (ctypes.c_int * nh)(*[d.shape[1] for d in douts])
    acts = (ctypes.c_int * nh)(*[_ACT[h[4]] for h in heads])
    scales = (ctypes.c_float * nh)(*[float(h[5]) for h in heads])
    karr = (ctypes.c_void_p * (5 * nh))(*[t.data_ptr() for h in heads for t in h[2]])
    barr = (ctypes.c_void_p * (5 * nh))(*[t.data_ptr() for h in heads for t in h[3]])
    check(lib.nfx_mlp128_bwd_heads(in_kind, _ptr(xyz), _ptr(xyz_dir), n, xyz_scale, _ptr(lxyz), nl, nh, blobs, dims, acts, scales,
                                   darr, _ptr(ws), ws.numel() * 2, karr, barr, _PREC[prec], _stream()), 'nfx_mlp128_bwd_heads')
    return ws


# ------------------------------------------------------------------------------ NeRF training ops
def pack_nerf_train_weights(kernels, biases, prec='bf16'):
    """Train blob (forward + dgrad fragments) of one NeRF network, for nerf_mlp_bwd."""
    if len(kernels) != 12 or len(biases) != 12:
        raise _capi.NfxError("pack_nerf_train_weights: need 12 kernels and 12 biases")
    return _pack(lib.nfx_nerf_train_packed_bytes, lib.nfx_nerf_pack_train_weights, kernels, biases, (_PREC[prec],))


def composite_bwd(rgbs, z, rayd, d_rgb, white_bg=True, noise=None):
    """d_rgbs[N,S,4] = dLoss/d rgbs given d_rgb[N,3] = dLoss/d (composited rgb)."""
    rgbs = _dev(rgbs, 'rgbs', (None, None, 4))
    n, s = rgbs.shape[:2]
    z = _dev(z, 'z', (n, s))
    rayd = _dev(rayd, 'rayd', (n, 3))
    noise = _dev(noise, 'noise', (n, s))
    d_rgb = _dev(d_rgb, 'd_rgb', (n, 3))
    out = torch.empty_like(rgbs)
    check(lib.nfx_composite_bwd(_ptr(rgbs), _ptr(z), _ptr(rayd), _ptr(noise), n, s, int(white_bg), _ptr(d_rgb),
                                _ptr(out), _stream()), 'nfx_composite_bwd')
    return out


NERF_BWD_MAX_POINTS = 1 << 19   # ~5 GB of feature-major workspace per call; more rays are processed in slices


NERF_BWD_STATS = None   # a list: nerf_mlp_bwd appends (points with a gradient [device int32 copy], points) per call (bench.py)


def nerf_bwd_list_words(n_pts):
    """Word offsets of (count, per-1024-point counts, indices) and the total of the row list at the END of the workspace
    of nfx_nerf_mlp_bwd (nerf_bwd.hip:nfx_nerf_bwd_list_bytes)."""
    nb = (n_pts + 1023) // 1024
    b0 = 4 + (nb + 3) // 4 * 4
    return 0, 4, b0, b0 + (n_pts + 3) // 4 * 4


def nerf_mlp_bwd(rayo, rayd, z, d_rgbs, blob, dkernels, dbiases, prec='bf16'):
    """Accumulate the weight gradients of one nerf_mlp_fwd call into `dkernels` / `dbiases` (lists of 12 fp32
    CUDA tensors, Keras layout) given d_rgbs[N,S,4].  Only the points whose d_rgbs is not four zeros are differentiated
    (the library builds their list on the device; option nerf_bwd_rows = 0: every point)."""
    rayo = _dev(rayo, 'rayo', (None, 3))
    n = rayo.shape[0]
    rayd = _dev(rayd, 'rayd', (n, 3))
    z = _dev(z, 'z', (n, None))
    s = z.shape[1]
    d_rgbs = _dev(d_rgbs, 'd_rgbs', (n, s, 4))
    for t in list(dkernels) + list(dbiases):
        _dev(t, 'gradient buffer')
    karr = (ctypes.c_void_p * 12)(*[t.data_ptr() for t in dkernels])
    barr = (ctypes.c_void_p * 12)(*[t.data_ptr() for t in dbiases])
    rays_per_call = max(1, NERF_BWD_MAX_POINTS // s)
    ws = None
    for lo in range(0, n, rays_per_call):
        hi = min(n, lo + rays_per_call)
        ws_bytes = lib.nfx_nerf_bwd_workspace_bytes(hi - lo, s)
        if ws is None or ws.numel() * 2 < ws_bytes:
            ws = torch.empty((max(ws_bytes, 16) // 2,), dtype=torch.bfloat16, device=z.device)
        check(lib.nfx_nerf_mlp_bwd(_ptr(rayo[lo:hi]), _ptr(rayd[lo:hi]), _ptr(z[lo:hi]), hi - lo, s, _ptr(blob),
                                   _PREC[prec], _ptr(d_rgbs[lo:hi]), _ptr(ws), ws.numel() * 2, karr, barr,
                                   _stream()), 'nfx_nerf_mlp_bwd')
        if NERF_BWD_STATS is not None:
            first = ws_bytes // 4 - nerf_bwd_list_words((hi - lo) * s)[3]
            NERF_BWD_STATS.append((ws.view(torch.int32)[first:first + 1].clone(), (hi - lo) * s))
    return ws


# ------------------------------------------------------------------------------ geometry extraction ops
def pack_nerf_geom_weights(kernels, biases, prec='bf16'):
    """Blob of nerf_sigma_grad: encoder + sigma tile, transposed encoder, input-gradient tiles."""
    if len(kernels) != 12 or len(biases) != 12:
        raise _capi.NfxError("pack_nerf_geom_weights: need 12 kernels and 12 biases")
    return _pack(lib.nfx_nerf_geom_packed_bytes, lib.nfx_nerf_pack_geom_weights, kernels, biases, (_PREC[prec],))


def _ray_args(rayo, rayd, z, blob):
    rayo = _dev(rayo, 'rayo', (None, 3))
    n = rayo.shape[0]
    rayd = _dev(rayd, 'rayd', (n, 3))
    z = _dev(z, 'z', (n, None))
    if not blob.is_cuda or blob.dtype != torch.uint8:
        raise _capi.NfxError("blob must be a CUDA uint8 tensor")
    return rayo, rayd, z, n, z.shape[1]


def nerf_sigma_fwd(rayo, rayd, z, blob, prec='bf16'):
    """sigma_raw[N,S] (no relu) at rayo + rayd*z from the GEOM blob (pack_nerf_geom_weights); the rgb head is neither
    evaluated nor streamed."""
    rayo, rayd, z, n, s = _ray_args(rayo, rayd, z, blob)
    out = torch.empty((n, s), dtype=torch.float32, device=z.device)
    check(lib.nfx_nerf_sigma_fwd(_ptr(rayo), _ptr(rayd), _ptr(z), n, s, _ptr(blob), _PREC[prec], _ptr(out),
                                 _stream()), 'nfx_nerf_sigma_fwd')
    return out


def nerf_refine_last_sample(rayo, rayd, z, rgbs, geom_blob_fp32):
    """Overwrites the density of every ray's LAST sample in rgbs[N,S,4] (in place) with the fp32-class density kernel's
    value (nerf_geom_x3.hip through nfx_nerf_sigma_fwd, NFX_PREC_FP32).  That sample gets dist = 1e10 when compositing
    (nerf.py:186-191): alpha_last = [sigma_last > 0] exactly, so its SIGN is the only bit of the ray that a bf16 kernel
    can get wrong by a whole pixel value; 1 / S of the points at ~3x the cost."""
    n, s = z.shape
    if n == 0 or s < 2:
        return rgbs
    sig = nerf_sigma_fwd(rayo, rayd, z[:, -1:].contiguous(), geom_blob_fp32, 'fp32')
    rgbs[:, -1, 3] = sig[:, 0]
    return rgbs


# The selection of nfx_nerf_refine_select as the render uses it, calibrated on all 640 000 rays of a view of a fitted NeRF
# against the fp32-class render (scripts/coarse_refine_residual.py, profiles/r06/coarse_refine_residual*.json): 2195 rays above
# 3e-2 with the plain bf16 coarse pass; 0 (max 1.6e-2 = the fp32-class coarse pass's own figure) with T > 1e-2, alpha in
# (1e-2, 0.99), no dilation, |sigma| < 0.3 — 6.9 % of the coarse samples — and with every wider rule; 767 with alpha in
# (0.05, 0.95); 27 without the |sigma| rule however wide the rest.  Shipped: one notch wider than the cheapest rule that held.
REFINE_T_MIN, REFINE_A_LO, REFINE_A_HI, REFINE_DILATE = 1e-3, 1e-3, 0.999, 0
REFINE_MARGIN_FACTOR = 2.       # sigma_margin = factor x the measured 99.9 % quantile of |sigma_bf16 - sigma_fp32class|


def nerf_refine_coarse(rayo, rayd, z, rgbs, geom_blob_fp32, t_min=REFINE_T_MIN, a_lo=REFINE_A_LO, a_hi=REFINE_A_HI,
                       dilate=REFINE_DILATE, sigma_margin=0., want_count=False):
    """Overwrites, in place, the density in rgbs[N,S,4] of the coarse samples that decide where the inverse-CDF sampler
    (util/math.py:71-94) puts the fine samples with the fp32-class density kernel's value: samples that are visible
    (T_i > t_min) and either not saturated (a_lo < alpha_i < a_hi) or undecided (|sigma_i| < sigma_margin: which side of
    the relu the sample is on lies within the bf16 kernel's error — on a near-miss ray, whose weights sum to almost nothing,
    one such sample is the whole pdf), and `dilate` neighbours either way.  Two launches
    (nfx_nerf_refine_select, nfx_nerf_sigma_refine); the list and its length stay on the device.  Returns rgbs (and the
    1-element int32 count tensor if `want_count`)."""
    rayo, rayd, z, n, s = _ray_args(rayo, rayd, z, geom_blob_fp32)
    if not (isinstance(rgbs, torch.Tensor) and rgbs.is_contiguous()):
        raise _capi.NfxError("nerf_refine_coarse: rgbs is updated in place and must be contiguous")
    rgbs = _dev(rgbs, 'rgbs', (n, s, 4))
    count = torch.empty(1, dtype=torch.int32, device=z.device)
    if n == 0 or s < 2:
        count.zero_()
        return (rgbs, count) if want_count else rgbs
    lst = torch.empty(n * s, dtype=torch.int32, device=z.device)
    check(lib.nfx_nerf_refine_select(_ptr(rgbs), _ptr(z), _ptr(rayd), n, s, t_min, a_lo, a_hi, sigma_margin, dilate, _ptr(lst), _ptr(count),
                                     _stream()), 'nfx_nerf_refine_select')
    check(lib.nfx_nerf_sigma_refine(_ptr(rayo), _ptr(rayd), _ptr(z), n, s, _ptr(geom_blob_fp32), _ptr(lst), _ptr(count),
                                    _ptr(rgbs), _stream()), 'nfx_nerf_sigma_refine')
    return (rgbs, count) if want_count else rgbs


def nerf_coarse_error(rayo, rayd, z, rgbs, geom_blob_fp32, max_rays=4096, q=0.999, geom_blob_bf16=None):
    """(alpha_error, sigma_error) of the bf16 coarse pass for THESE weights: sigma_error = the q-quantile of
    |sigma_bf16 - sigma_fp32class| over the samples of up to `max_rays` rays of the batch; alpha_error = that times the mean
    sample spacing = how far the bf16 error can move a sample's alpha.  The render's `coarse_precision = auto` measures this
    once per weight version (one host read): the selective refinement is on when alpha_error exceeds `coarse_refine_gate`
    (measured: glorot "opaque" weights of the bench 1.5e-3, a NeRF fitted to a scene 1.0e-2), and sigma_error sizes the
    |sigma| < margin rule of nerf_refine_coarse.  `rgbs` = the bf16 pass's output for these rays, or None: then the bf16
    densities of the picked rays come from the bf16 density kernel (`geom_blob_bf16`; bit-identical to the MLP kernel's sigma)."""
    n = min(int(z.shape[0]), max_rays)
    if n == 0 or z.shape[1] < 2:
        return 0., 0.
    step = max(1, int(z.shape[0]) // n)
    pick = slice(0, n * step, step)
    o, d, zz = rayo[pick].contiguous(), rayd[pick].contiguous(), z[pick].contiguous()
    s32 = nerf_sigma_fwd(o, d, zz, geom_blob_fp32, 'fp32')
    s16 = rgbs[pick][..., 3] if rgbs is not None else nerf_sigma_fwd(o, d, zz, geom_blob_bf16, 'bf16')
    err = (s16 - s32).abs()[:, :-1].reshape(-1)
    spacing = ((zz[:, 1:] - zz[:, :-1]).mean() * d.norm(dim=1).mean())
    sig = torch.quantile(err[:4000000], q)
    both = torch.stack((sig * spacing, sig)).tolist()
    return float(both[0]), float(both[1])


SIGMA_GRAD_STATS = None   # a list: nerf_sigma_grad appends (samples with a density [device int32 copy], samples) per call (bench.py)


def nerf_sigma_grad(rayo, rayd, z, geom_blob, prec='bf16'):
    """(normal[N,S,3], sigma_raw[N,S]) with normal = -l2_normalize(d relu(sigma)/dx)."""
    rayo, rayd, z, n, s = _ray_args(rayo, rayd, z, geom_blob)
    out = torch.empty((n, s, 4), dtype=torch.float32, device=z.device)
    if _capi.get_option("sigma_grad_rows") != 0 and n * s < 2 ** 31:
        # the reverse sweep only over the samples with a positive density (every other sample's normal is zero: relu has no
        # slope there) — the same values, and most samples of a fitted scene are empty space
        ws_bytes = lib.nfx_nerf_sigma_grad_workspace_bytes(n, s)
        ws = torch.empty((max(ws_bytes, 16),), dtype=torch.uint8, device=z.device)
        check(lib.nfx_nerf_sigma_grad_rows(_ptr(rayo), _ptr(rayd), _ptr(z), n, s, _ptr(geom_blob), _PREC[prec], _ptr(out),
                                           _ptr(ws), ws.numel(), _stream()), 'nfx_nerf_sigma_grad_rows')
        if SIGMA_GRAD_STATS is not None:
            first = (n * s * 4 + 15) // 16 * 4
            SIGMA_GRAD_STATS.append((ws.view(torch.int32)[first:first + 1].clone(), n * s))
    else:
        check(lib.nfx_nerf_sigma_grad(_ptr(rayo), _ptr(rayd), _ptr(z), n, s, _ptr(geom_blob), _PREC[prec], _ptr(out),
                                      _stream()), 'nfx_nerf_sigma_grad')
    return out[..., :3], out[..., 3]

def amsgrad_step(p, g, m, v, vhat, lr, step, beta1=0.9, beta2=0.999, eps=1e-7):
    """In-place Keras Adam(amsgrad=True) update of the flat fp32 buffer `p` (step is 1-based)."""
    for t in (p, g, m, v, vhat):
        _dev(t, 'optimizer buffer')
    check(lib.nfx_amsgrad_step(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(vhat), p.numel(), lr, beta1, beta2,
                               eps, step, _stream()), 'nfx_amsgrad_step')


def amsgrad_step_size(lr, step, beta1=0.9, beta2=0.999):
    """lr_t = lr * sqrt(1 - beta2^step) / (1 - beta1^step), evaluated as the library does (double, rounded once)."""
    return float(lib.nfx_amsgrad_step_size(lr, beta1, beta2, step))


def amsgrad_step_dev(p, g, m, v, vhat, lr_t_dev, beta1=0.9, beta2=0.999, eps=1e-7):
    """amsgrad_step with the step size read from the 1-element CUDA tensor `lr_t_dev` (hipGraph-replayable)."""
    for t in (p, g, m, v, vhat, lr_t_dev):
        _dev(t, 'optimizer buffer')
    check(lib.nfx_amsgrad_step_dev(_ptr(p), _ptr(g), _ptr(m), _ptr(v), _ptr(vhat), p.numel(), _ptr(lr_t_dev), beta1,
                                   beta2, eps, _stream()), 'nfx_amsgrad_step_dev')


def shade_bwd(xyz, cam, normal, albedo, lvis, lxyz, lareas, light, drgb, d_light, rough=None, spec=None,
              spec_scale=1., f0=0.04, linear2srgb=True):
    """Backward of shade_fwd for one light [L,3].  Returns (d_albedo, d_normal, d_lvis, d_rough|d_spec);
    adds the light's gradient to d_light [L,3] (order-independent fixed-point sum: bit-reproducible)."""
    xyz, cam, normal, albedo, rough, spec, lvis, lxyz, lareas, n, nl = _shade_common(
        xyz, cam, normal, albedo, rough, spec, lvis, lxyz, lareas)
    light = _dev(light, 'light', (nl, 3))
    drgb = _dev(drgb, 'drgb', (n, 3))
    d_light = _dev(d_light, 'd_light', (nl, 3))
    dev_ = xyz.device
    d_albedo = torch.empty((n, 3), dtype=torch.float32, device=dev_)
    d_normal = torch.empty((n, 3), dtype=torch.float32, device=dev_)
    d_lvis = torch.empty((n, nl), dtype=torch.float32, device=dev_)
    d_rough = torch.empty((n,), dtype=torch.float32, device=dev_) if spec is None else None
    d_spec = torch.empty((n, nl), dtype=torch.float32, device=dev_) if spec is not None else None
    ws = torch.empty((max(lib.nfx_shade_bwd_workspace_bytes(nl), 8) // 8,), dtype=torch.int64, device=dev_)
    check(lib.nfx_shade_bwd(_ptr(xyz), _ptr(cam), _ptr(normal), _ptr(albedo), _ptr(rough), _ptr(spec), spec_scale,
                            f0, _ptr(lvis), _ptr(lxyz), _ptr(lareas), _ptr(light), n, nl, int(linear2srgb),
                            _ptr(drgb), _ptr(d_albedo), _ptr(d_rough), _ptr(d_spec), _ptr(d_normal), _ptr(d_lvis),
                            _ptr(d_light), _ptr(ws), ws.numel() * 8, _stream()), 'nfx_shade_bwd')
    return d_albedo, d_normal, d_lvis, (d_rough if spec is None else d_spec)


def pack_brdf_train_weights(kernels, biases, z_dim, prec='bf16'):
    """Train blob of the (frozen) learned-BRDF MLP: forward + dgrad + input-gradient fragments."""
    return _pack(lambda zd, p: lib.nfx_brdf_train_packed_bytes(), lib.nfx_brdf_pack_train_weights, kernels,
                 biases, (z_dim, _PREC[prec]))


def brdf_spec_bwd(xyz, cam, normal, z, lxyz, blob, dspec, prec='bf16'):
    """(d_z [n, z_dim], d_normal [n, 3]) from dspec [n, L] through the frozen BRDF prior."""
    xyz = _dev(xyz, 'xyz', (None, 3))
    n = xyz.shape[0]
    z = _dev(z, 'z', (n, None))
    lxyz = _dev(lxyz, 'lxyz', (None, 3))
    nl = lxyz.shape[0]
    dspec = _dev(dspec, 'dspec', (n, nl))
    d_z = torch.zeros_like(z)
    d_normal = torch.zeros((n, 3), dtype=torch.float32, device=xyz.device)
    ws = torch.empty((max(lib.nfx_brdf_spec_bwd_workspace_bytes(z.shape[1], n), 8) // 8,), dtype=torch.int64,
                     device=xyz.device)
    # round 6: only the rows with a non-zero upstream gradient are re-computed and differentiated (the shading backward zeroes
    # d spec of every back-facing light: half of the rows); option brdf_bwd_rows = 0 keeps every row (same bits)
    list_bytes = lib.nfx_brdf_spec_bwd_list_bytes(n, nl) if _capi.get_option("brdf_bwd_rows") != 0 else 0
    lws = torch.empty((list_bytes // 4,), dtype=torch.int32, device=xyz.device) if list_bytes else None
    check(lib.nfx_brdf_spec_bwd_rows(_ptr(xyz), _ptr(_dev(cam, 'cam', (n, 3))), _ptr(_dev(normal, 'normal', (n, 3))),
                                     _ptr(z), z.shape[1], _ptr(lxyz), nl, _ptr(blob), _PREC[prec], n, _ptr(dspec),
                                     _ptr(d_z), _ptr(d_normal), _ptr(ws), ws.numel() * 8,
                                     None if lws is None else ctypes.c_void_p(lws.data_ptr()), list_bytes, _stream()),
          'nfx_brdf_spec_bwd')
    return d_z, d_normal


def brdf_rows_fwd(z, rusink, blob, reci=True, prec='bf16'):
    """The BRDF prior on explicit rows: out[n] (reci=False) or out[2n] (rows n.. = the same inputs at phi_d + pi) =
    softplus(out(mlp([z, posenc2(rusink)]))).  `blob` = pack_brdf_train_weights(...)."""
    z = _dev(z, 'z', (None, None))
    n = z.shape[0]
    rusink = _dev(rusink, 'rusink', (n, 3))
    out = torch.empty(((2 if reci else 1) * n,), dtype=torch.float32, device=z.device)
    check(lib.nfx_brdf_rows_fwd(_ptr(z), z.shape[1], _ptr(rusink), n, int(bool(reci)), _ptr(blob), _PREC[prec],
                                _ptr(out), _stream()), 'nfx_brdf_rows_fwd')
    return out


def brdf_rows_bwd(z, rusink, blob, dout, dkernels, dbiases, reci=True, prec='bf16'):
    """d_z[rows, z_dim] (per row: sum the two halves for the gradient of z[n]) from dout[rows]; ACCUMULATES the weight
    gradients into `dkernels` / `dbiases` (5 fp32 CUDA tensors each, Keras layout)."""
    z = _dev(z, 'z', (None, None))
    n, zd = z.shape
    rows = (2 if reci else 1) * n
    rusink = _dev(rusink, 'rusink', (n, 3))
    dout = _dev(dout.reshape(rows), 'dout', (rows,))
    for t in list(dkernels) + list(dbiases):
        _dev(t, 'gradient buffer')
    d_z = torch.empty((rows, zd), dtype=torch.float32, device=z.device)
    ws_bytes = lib.nfx_brdf_rows_bwd_workspace_bytes(zd, n, int(bool(reci)))
    ws = torch.empty((max(ws_bytes, 16) // 2,), dtype=torch.bfloat16, device=z.device)
    karr = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in dkernels])
    barr = (ctypes.c_void_p * 5)(*[t.data_ptr() for t in dbiases])
    check(lib.nfx_brdf_rows_bwd(_ptr(z), zd, _ptr(rusink), n, int(bool(reci)), _ptr(blob), _PREC[prec], _ptr(dout),
                                _ptr(ws), ws.numel() * 2, _ptr(d_z), karr, barr, _stream()), 'nfx_brdf_rows_bwd')
    return d_z


# ------------------------------------------------------------------------------ fused per-ray losses
class _LossTerm(ctypes.Structure):   # include/nfx.h: nfx_loss_term
    _fields_ = [('a', ctypes.c_void_p), ('b', ctypes.c_void_p), ('ga', ctypes.c_void_p), ('gb', ctypes.c_void_p),
                ('d', ctypes.c_int), ('w', ctypes.c_float), ('kind', ctypes.c_int), ('flags', ctypes.c_int)]


LOSS_KIND = {'mse': 0, 'mae': 1}
LOSS_BLEND_A, LOSS_BLEND_B, LOSS_ACCUM_A, LOSS_ACCUM_B = 1, 2, 4, 8


def _loss_table(terms, grads=None):
    """terms: [(a, b, weight, kind, blend_a, blend_b)]; grads: per term (ga, gb, accum_a, accum_b) or None."""
    if not 1 <= len(terms) <= 8:
        raise _capi.NfxError("pair_loss: 1..8 terms")
    n = terms[0][0].shape[0]
    table = (_LossTerm * len(terms))()
    for i, (a, b, w, kind, blend_a, blend_b) in enumerate(terms):
        a = _dev(a, 'loss operand', (n, None))
        b = _dev(b, 'loss operand', (n, a.shape[1]))
        flags = (LOSS_BLEND_A if blend_a else 0) | (LOSS_BLEND_B if blend_b else 0)
        ga = gb = None
        if grads is not None:
            ga, gb, acc_a, acc_b = grads[i]
            flags |= (LOSS_ACCUM_A if acc_a else 0) | (LOSS_ACCUM_B if acc_b else 0)
        table[i] = _LossTerm(a.data_ptr(), b.data_ptr(), ga.data_ptr() if ga is not None else None,
                             gb.data_ptr() if gb is not None else None, a.shape[1], float(w), LOSS_KIND[kind], flags)
    return table, n


def pair_loss_fwd(terms, alpha=None, bg=0.):
    """loss[n] = sum_t w_t mean_d f_t(A_t - B_t) with optional alpha blending of A / B onto `bg` (nfx.h)."""
    table, n = _loss_table(terms)
    dev = terms[0][0].device
    alpha = None if alpha is None else _dev(alpha.reshape(n), 'alpha', (n,))
    loss = torch.empty((n,), dtype=torch.float32, device=dev)
    check(lib.nfx_pair_loss_fwd(table, len(terms), _ptr(alpha), float(bg), n, _ptr(loss), _stream()),
          'nfx_pair_loss_fwd')
    return loss


def pair_loss_bwd(terms, grads, dloss, alpha=None, bg=0.):
    """Writes / accumulates d loss / d A, d loss / d B into the buffers of `grads` (see _loss_table)."""
    table, n = _loss_table(terms, grads)
    alpha = None if alpha is None else _dev(alpha.reshape(n), 'alpha', (n,))
    dloss = _dev(dloss.reshape(n), 'dloss', (n,))
    check(lib.nfx_pair_loss_bwd(table, len(terms), _ptr(alpha), float(bg), n, _ptr(dloss), _stream()),
          'nfx_pair_loss_bwd')
