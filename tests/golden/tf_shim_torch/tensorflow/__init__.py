"""torch-CPU stand-in for the part of the TensorFlow 2 API the google/nerfactor TRAINING path calls — the
reverse-mode sibling of ../../tf_shim (NumPy, forward only).  Tensors are float32 `torch.Tensor`s, so
`tf.GradientTape().gradient(...)` is torch's reverse-mode autodiff THROUGH THE REFERENCE'S OWN PYTHON
(nerfactor/models/*.py, trainvali.py:273-285), `tf.custom_gradient` keeps the backward function the reference wrote
(util/math.py:24-60) and `tf.keras.optimizers.Adam(amsgrad=True)` is the TF 2.2 dense update rule.  Where TensorFlow's
registered gradient differs from torch's at a non-differentiable point the TF rule is implemented here (tf.maximum /
tf.minimum ties go to the FIRST argument, tf.math.divide_no_nan has zero gradients where the denominator is 0).
Test infrastructure only (tests/golden/make_reference_grad_golden.py).
"""
import numpy as _np
import torch as _torch

float32, float64, int32, int64, uint8 = _torch.float32, _torch.float64, _torch.int32, _torch.int64, _torch.uint8
bool = _torch.bool  # noqa: A001
newaxis = None
Tensor = _torch.Tensor
__version__ = '2.3-torch-shim'
_torch.set_default_dtype(_torch.float32)


def _t(x, dtype=None):
    """tf.convert_to_tensor's dtype inference: Python float / float64 array -> float32, ints -> int32."""
    if isinstance(x, _torch.Tensor):
        return x if dtype is None or x.dtype == dtype else x.to(dtype)
    if isinstance(x, _Shape):
        x = list(x)
    a = _np.asarray(x)
    if a.dtype == _np.float64:
        a = a.astype(_np.float32)
    elif a.dtype == _np.int64:
        a = a.astype(_np.int32)
    if a.dtype.kind in 'SUO':
        return a
    out = _torch.from_numpy(_np.ascontiguousarray(a))
    return out if dtype is None else out.to(dtype)


class _Shape(tuple):
    """tf.shape(x) / x.get_shape(): ints that index, slice and unpack like the 1-D int32 tensor TF returns."""
    def as_list(self):
        return list(self)

    def __getitem__(self, i):
        r = tuple.__getitem__(self, i)
        return _Shape(r) if isinstance(i, slice) else r


def as_dtype(d):
    return d


def convert_to_tensor(value, dtype=None, **_):
    return _t(value, dtype)


constant = convert_to_tensor


def is_tensor(x):
    return isinstance(x, _torch.Tensor)


def Variable(initial_value, trainable=True, **_):
    v = _t(initial_value).detach().clone().requires_grad_(builtins_bool(trainable))
    v.trainable = builtins_bool(trainable)
    return v


import builtins as _b  # noqa: E402
builtins_bool = _b.bool
_range = _b.range
_abs = _b.abs


def identity(x):
    return _t(x)


def stop_gradient(x):
    return _t(x).detach()


def ensure_shape(x, shape):
    assert len(x.shape) == len(shape) and all(s is None or s == d for s, d in zip(shape, x.shape)), (x.shape, shape)
    return x


def cast(x, dtype):
    return _t(x).to(dtype)


# ------------------------------------------------------------------ shapes
def shape(x):
    return _torch.tensor([int(s) for s in _t(x).shape], dtype=int32)


def rank(x):
    return _t(x).dim()


def _ints(shape_):
    return tuple(int(s) for s in shape_)


def reshape(x, shape):  # noqa: A002
    return _t(x).reshape(_ints(shape))


def transpose(x, perm=None):
    x = _t(x)
    return x.permute(*perm) if perm is not None else x.permute(*reversed(_range(x.dim())))


def expand_dims(x, axis):
    return _t(x).unsqueeze(axis)


def concat(values, axis):
    return _torch.cat([_t(v) for v in values], dim=axis)


def stack(values, axis=0):
    return _torch.stack([_t(v) for v in values], dim=axis)


def tile(x, multiples):
    return _t(x).repeat(*_ints(multiples))


def broadcast_to(x, shape):  # noqa: A002
    return _t(x).expand(*_ints(shape)).clone()


def zeros(shape, dtype=float32):  # noqa: A002
    return _torch.zeros(_ints(shape), dtype=dtype)


def ones(shape, dtype=float32):  # noqa: A002
    return _torch.ones(_ints(shape), dtype=dtype)


def zeros_like(x):
    return _torch.zeros_like(_t(x))


def ones_like(x):
    return _torch.ones_like(_t(x))


def linspace(start, stop, num):
    num = int(num)
    if num == 1:
        return _t(_np.asarray([start], _np.float32))
    step = (_np.float32(stop) - _np.float32(start)) / _np.float32(num - 1)
    out = _np.float32(start) + _np.arange(num, dtype=_np.float32) * step
    out[-1] = _np.float32(stop)
    return _t(out)


def range(start, limit=None, delta=1, dtype=None):  # noqa: A001
    if limit is None:
        start, limit = 0, start
    a = _np.arange(int(start), int(limit), int(delta)).astype(_np.int32)
    return _t(a, dtype)


def meshgrid(*args, indexing='xy'):
    return list(_torch.meshgrid(*[_t(a) for a in args], indexing=indexing))


def roll(x, shift, axis):
    return _torch.roll(_t(x), shift, axis)


# ------------------------------------------------------------------ elementwise / reductions
def _un(f):
    return lambda x, *a, **k: f(_t(x), *a, **k)


exp, sqrt, square, abs, sin, cos, acos = (_un(f) for f in (  # noqa: A001
    _torch.exp, _torch.sqrt, _torch.square, _torch.abs, _torch.sin, _torch.cos, _torch.acos))


def atan2(y, x):
    return _torch.atan2(_t(y), _t(x))


def rsqrt(x):
    return 1. / _torch.sqrt(_t(x))


def _like(a, b):
    a = a if isinstance(a, _torch.Tensor) else _t(_np.float32(a) if isinstance(a, float) else a)
    b = b if isinstance(b, _torch.Tensor) else _t(_np.float32(b) if isinstance(b, float) else b)
    return a, b


def maximum(a, b):
    """MaximumGrad of TensorFlow: the whole gradient goes to `a` where a >= b (torch splits ties)."""
    a, b = _like(a, b)
    a, b = _torch.broadcast_tensors(a, b)
    return _torch.where(a >= b, a, b)


def minimum(a, b):
    a, b = _like(a, b)
    a, b = _torch.broadcast_tensors(a, b)
    return _torch.where(a <= b, a, b)


def multiply(a, b):
    return _t(a) * _t(b)


def equal(a, b):
    if isinstance(a, _torch.Tensor) or isinstance(b, _torch.Tensor):
        return _t(a) == _t(b)
    return a == b


def logical_or(a, b):
    return a | b if isinstance(a, _torch.Tensor) else (a or b)


def logical_and(a, b):
    return a & b if isinstance(a, _torch.Tensor) else (a and b)


def clip_by_value(x, clip_value_min, clip_value_max):
    x = _t(x)
    lo = None if clip_value_min in (-_np.inf,) else float(clip_value_min)
    hi = None if clip_value_max in (_np.inf,) else float(clip_value_max)
    return _torch.clamp(x, min=lo, max=hi)      # gradient 1 on [min, max] inclusive, as TF's ClipByValueGrad


def _axis(axis):
    return tuple(axis) if isinstance(axis, (list, tuple)) else axis


def reduce_sum(x, axis=None, keepdims=False):
    x = _t(x)
    if axis is not None and _axis(axis) == ():      # TF: an empty axis list reduces nothing (torch: everything)
        return x
    return x.sum() if axis is None else x.sum(dim=_axis(axis), keepdim=keepdims)


def reduce_mean(x, axis=None, keepdims=False):
    x = _t(x)
    if axis is not None and _axis(axis) == ():
        return x
    return x.mean() if axis is None else x.mean(dim=_axis(axis), keepdim=keepdims)


def reduce_min(x, axis=None):
    x = _t(x)
    return x.min() if axis is None else x.min(dim=axis).values


def reduce_max(x, axis=None):
    x = _t(x)
    return x.max() if axis is None else x.max(dim=axis).values


def cumsum(x, axis=0):
    return _torch.cumsum(_t(x), dim=axis)


def matmul(a, b):
    return _torch.matmul(_t(a), _t(b))


def einsum(eq, *ops):
    return _torch.einsum(eq, *[_t(o) for o in ops])


def sort(x, axis=-1):
    return _torch.sort(_t(x), dim=axis).values


def where(condition, x=None, y=None):
    if x is None:
        return _torch.nonzero(_t(condition))
    x, y = _like(x, y)
    return _torch.where(_t(condition), x, y)


def boolean_mask(tensor, mask, axis=None):
    assert axis in (None, 0)
    return _t(tensor)[_t(mask).to(_torch.bool)]


def gather(params, indices, axis=0, batch_dims=0):
    params, indices = _t(params), _t(indices).long()
    if batch_dims == 0:
        return _torch.index_select(params, axis, indices.reshape(-1)).reshape(
            params.shape[:axis] + indices.shape + params.shape[axis + 1:])
    # the one batched form the reference uses: gather along the last axis with all leading axes batched
    assert axis in (-1, params.dim() - 1) and batch_dims == params.dim() - 1, (axis, batch_dims, params.shape)
    flat_p = params.reshape(-1, params.shape[-1])
    flat_i = indices.reshape(flat_p.shape[0], -1)
    return _torch.gather(flat_p, 1, flat_i).reshape(indices.shape)


def _nd_index(indices):
    indices = _t(indices).long()
    return tuple(indices[..., k] for k in _range(indices.shape[-1]))


def gather_nd(params, indices):
    return _t(params)[_nd_index(indices)]


def scatter_nd(indices, updates, shape):  # noqa: A002
    updates = _t(updates)
    out = _torch.zeros(_ints(shape), dtype=updates.dtype)
    return out.index_put(_nd_index(indices), updates, accumulate=True)


def tensor_scatter_nd_update(tensor, indices, updates):
    return _t(tensor).index_put(_nd_index(indices), _t(updates), accumulate=False)


def searchsorted(sorted_sequence, values, side='left'):
    return _torch.searchsorted(_t(sorted_sequence).contiguous(), _t(values).contiguous(), right=(side == 'right')).to(int32)


def cond(pred, true_fn, false_fn):
    return true_fn() if builtins_bool(pred) else false_fn()


class control_dependencies:  # noqa: N801
    def __init__(self, deps):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def custom_gradient(f):
    """tf.custom_gradient: f(*args) -> (value, grad_fn); grad_fn(dy) -> gradient per positional argument.  The forward
    runs without recording, the backward is the reference's own grad_fn."""
    class _Fn(_torch.autograd.Function):
        @staticmethod
        def forward(ctx, *args):
            with _torch.no_grad():
                value, grad_fn = f(*[a.detach() if isinstance(a, _torch.Tensor) else a for a in args])
            ctx.grad_fn_ref = grad_fn
            ctx.n = len(args)
            return value

        @staticmethod
        def backward(ctx, dy):
            with _torch.no_grad():
                g = ctx.grad_fn_ref(dy)
            g = list(g) if isinstance(g, (tuple, list)) else [g]
            return tuple(g + [None] * (ctx.n - len(g)))

    def wrapped(*args, **kwargs):
        assert not kwargs, 'keyword arguments of a custom_gradient function are not differentiated'
        return _Fn.apply(*[_t(a) for a in args])
    wrapped.__name__ = getattr(f, '__name__', 'custom_gradient')
    return wrapped


class GradientTape:
    """Reverse mode: gradient(target, sources) = torch.autograd.grad (None for a source the target does not reach)."""
    def __init__(self, persistent=False):
        self.persistent = persistent

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def watch(self, x):
        x.requires_grad_(True)

    def gradient(self, target, sources):
        single = isinstance(sources, _torch.Tensor)
        srcs = [sources] if single else list(sources)
        g = _torch.autograd.grad(target, srcs, allow_unused=True, retain_graph=self.persistent)
        return g[0] if single else list(g)


def random_normal_initializer(mean=0., stddev=1.):
    def init(shape, dtype=float32):  # noqa: A002
        return _t((random._rng.standard_normal(tuple(shape)) * stddev + mean).astype(_np.float32))
    return init


# ------------------------------------------------------------------ namespaces
class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _cumprod(x, axis=0, exclusive=False):
    x = _t(x)
    out = _torch.cumprod(x, dim=axis)
    if exclusive:
        ones_ = _torch.ones_like(x.narrow(axis, 0, 1))
        out = _torch.cat((ones_, out.narrow(axis, 0, x.shape[axis] - 1)), dim=axis)
    return out


def _divide_no_nan(a, b):
    a, b = _like(a, b)
    a, b = _torch.broadcast_tensors(a, b)
    zero = b == 0
    return _torch.where(zero, _torch.zeros_like(a), a / _torch.where(zero, _torch.ones_like(b), b))


def _l2_normalize(x, axis=None, epsilon=1e-12):
    x = _t(x)
    sq = (x * x).sum(dim=axis, keepdim=True)
    return x * rsqrt(maximum(sq, _torch.tensor(epsilon, dtype=x.dtype)))


def _norm(x, ord='euclidean', axis=None, keepdims=False):  # noqa: A002
    x = _t(x)
    return _torch.sqrt((x * x).sum(dim=axis, keepdim=keepdims))


def _cross(a, b):
    return _torch.cross(_t(a), _t(b), dim=-1)


def _floormod(x, y):
    return _torch.remainder(_t(x), float(y))


def _pow(x, y):
    return _torch.pow(_t(x), float(y))


_softplus = _un(_torch.nn.functional.softplus)
_sigmoid = _un(_torch.sigmoid)
_relu = _un(_torch.relu)

math = _NS(sin=sin, cos=cos, log=_un(_torch.log), sigmoid=_sigmoid, cumprod=_cumprod, divide_no_nan=_divide_no_nan,
           floormod=_floormod, l2_normalize=_l2_normalize, minimum=minimum, maximum=maximum, pow=_pow, sqrt=sqrt,
           exp=exp, abs=abs, softplus=_softplus)
linalg = _NS(l2_normalize=_l2_normalize, norm=_norm, cross=_cross)


def _compute_average_loss(per_example_loss, sample_weight=None, global_batch_size=None):
    return _t(per_example_loss).sum() / float(global_batch_size)


nn = _NS(relu=_relu, sigmoid=_sigmoid, softplus=_softplus, compute_average_loss=_compute_average_loss)


class _Random:
    def __init__(self):
        self._rng = _np.random.default_rng(0)

    def set_seed(self, seed):
        self._rng = _np.random.default_rng(seed)

    def uniform(self, shape, minval=0., maxval=1., dtype=float32):  # noqa: A002
        x = self._rng.random(_ints(shape)) * (maxval - minval) + minval
        if dtype in (int32, int64):      # integer draws lie in [minval, maxval) (datasets/nerf.py:123)
            return _t(_np.floor(x).astype(_np.int32 if dtype == int32 else _np.int64))
        return _t(x.astype(_np.float32))

    def normal(self, shape, mean=0., stddev=1., dtype=float32):  # noqa: A002
        return _t((self._rng.standard_normal(_ints(shape)) * stddev + mean).astype(_np.float32))


random = _Random()


def _assert_greater(x, y, message=None):
    if not builtins_bool((_t(x).detach() > y).all()):
        raise AssertionError(message or 'assert_greater failed')


def _check_numerics(x, message):
    if not builtins_bool(_torch.isfinite(_t(x).detach()).all()):
        raise FloatingPointError(message)
    return x


def _debug_assert(condition, data=None, **_):
    if not builtins_bool(_t(condition).all() if isinstance(condition, _torch.Tensor) else condition):
        raise AssertionError(data)


debugging = _NS(assert_greater=_assert_greater, check_numerics=_check_numerics, Assert=_debug_assert)


def _resize(images, size, method='bilinear', antialias=False):
    images = _t(images)
    if _ints(size) == tuple(images.shape[-3:-1]):
        return images.to(float32)
    raise NotImplementedError('tf.image.resize to a different size is outside this shim')


image = _NS(resize=_resize)


class _Restore:
    def expect_partial(self):
        return self


class _Checkpoint:
    def __init__(self, **kw):
        self.kw = kw

    def restore(self, path):
        return _Restore()


train = _NS(Checkpoint=_Checkpoint)
data = _NS(experimental=_NS(AUTOTUNE=-1))
string = _np.str_

from . import keras  # noqa: E402,F401
