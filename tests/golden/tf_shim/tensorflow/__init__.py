"""NumPy stand-in for the part of the TensorFlow 2 API the google/nerfactor model code calls (see ../README.md).

Tensors are float32/int32 `numpy.ndarray`s (subclass `Tensor`, which adds `.numpy()`); Python floats become float32
and Python ints int32, as `tf.convert_to_tensor` would make them.  Each function follows the semantics TensorFlow
documents for it; nothing here is specific to the reference.  Test infrastructure only.
"""
import numpy as _np

float32 = _np.float32
float64 = _np.float64
int32 = _np.int32
int64 = _np.int64
bool = _np.bool_  # noqa: A001  (tf.bool)
uint8 = _np.uint8
newaxis = None
__version__ = '2.3-numpy-shim'


class Tensor(_np.ndarray):
    """float32/int32 array.  `_tangent` ([..., 3], forward-mode derivative with respect to the watched [N, 3] input of
    the same row) is set by GradientTape.watch and carried only by the operations listed in the GradientTape note."""
    _tangent = None

    def numpy(self):
        return _np.asarray(self)

    def get_shape(self):
        return _Shape(self.shape)

    def _lin(self, other, out, scale_self, scale_other):
        ts, to = self._tangent, getattr(other, '_tangent', None)
        if ts is None and to is None:
            return out
        t = 0
        if ts is not None:
            t = t + ts * scale_self
        if to is not None:
            t = t + to * scale_other
        out._tangent = _np.broadcast_to(t, out.shape + (3,)).astype(_np.float32)
        return out

    def __add__(self, o):
        return self._lin(o, _np.ndarray.__add__(self, o), 1, 1)

    __radd__ = __add__

    def __sub__(self, o):
        return self._lin(o, _np.ndarray.__sub__(self, o), 1, -1)

    def __neg__(self):
        return self._lin(None, _np.ndarray.__neg__(self), -1, 0)

    def __mul__(self, o):
        if getattr(o, '_tangent', None) is not None:
            raise NotImplementedError('product of two watched tensors')
        out = _np.ndarray.__mul__(self, o)
        if self._tangent is not None:
            out._tangent = (self._tangent * _np.asarray(o, _np.float32)[..., None]).astype(_np.float32)
        return out

    __rmul__ = __mul__


class _Shape(tuple):
    def as_list(self):
        return list(self)


def _t(x, dtype=None):
    """tf.convert_to_tensor's dtype inference: Python float -> float32, Python int -> int32."""
    if isinstance(x, Tensor) and (dtype is None or x.dtype == as_dtype(dtype)):
        return x                                   # keeps a forward-mode tangent attached
    if isinstance(x, _np.ndarray):
        a = x
    else:
        a = _np.asarray(x)
        if a.dtype == _np.float64:
            a = a.astype(_np.float32)
        elif a.dtype == _np.int64:
            a = a.astype(_np.int32)
    if dtype is not None:
        a = a.astype(as_dtype(dtype), copy=False)
    return a.view(Tensor)


def as_dtype(d):
    return _np.dtype('float32' if d == 'float32' else d)


def convert_to_tensor(value, dtype=None, **_):
    return _t(value, dtype)


constant = convert_to_tensor


def is_tensor(x):
    return isinstance(x, Tensor)


def Variable(initial_value, trainable=True, **_):
    v = _np.array(initial_value, copy=True)
    return _t(v)


def identity(x):
    return _t(x)


def stop_gradient(x):
    return _t(x)


def ensure_shape(x, shape):
    assert len(x.shape) == len(shape) and all(s is None or s == d for s, d in zip(shape, x.shape)), (x.shape, shape)
    return x


def cast(x, dtype):
    return _t(_np.asarray(x).astype(as_dtype(dtype)))


# ------------------------------------------------------------------ shapes
def shape(x):
    return _np.asarray(_np.shape(x), _np.int32).view(Tensor)


def rank(x):
    return _np.ndim(x)


def reshape(x, shape):  # noqa: A002
    return _t(_np.reshape(_t(x), tuple(int(s) for s in shape)))


def transpose(x, perm=None):
    return _t(_np.transpose(x, perm))


def expand_dims(x, axis):
    return _t(_np.expand_dims(x, axis))


def concat(values, axis):
    values = [_t(v) for v in values]
    out = _t(_np.concatenate(values, axis))
    if any(v._tangent is not None for v in values):
        ax = axis if axis >= 0 else out.ndim + axis
        out._tangent = _np.concatenate([v._tangent if v._tangent is not None else _np.zeros(v.shape + (3,), _np.float32)
                                        for v in values], ax)
    return out


def stack(values, axis=0):
    return _t(_np.stack([_t(v) for v in values], axis))


def tile(x, multiples):
    return _t(_np.tile(x, tuple(int(m) for m in multiples)))


def broadcast_to(x, shape):  # noqa: A002
    return _t(_np.broadcast_to(_t(x), tuple(int(s) for s in shape)).copy())


def zeros(shape, dtype=float32):  # noqa: A002
    return _t(_np.zeros(tuple(int(s) for s in shape), as_dtype(dtype)))


def ones(shape, dtype=float32):  # noqa: A002
    return _t(_np.ones(tuple(int(s) for s in shape), as_dtype(dtype)))


def zeros_like(x):
    return _t(_np.zeros_like(_t(x)))


def ones_like(x):
    return _t(_np.ones_like(_t(x)))


def linspace(start, stop, num):
    # tf.linspace on float32: start + i * (stop - start) / (num - 1), last element = stop
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
    a = _np.arange(start, limit, delta)
    if dtype is None:
        a = a.astype(_np.int32) if a.dtype.kind == 'i' else a.astype(_np.float32)
    return _t(a, dtype)


def meshgrid(*args, indexing='xy'):
    return [_t(g) for g in _np.meshgrid(*args, indexing=indexing)]


def roll(x, shift, axis):
    return _t(_np.roll(x, shift, axis))


# ------------------------------------------------------------------ elementwise / reductions
def _un(f):
    return lambda x, *a, **k: _t(f(_t(x), *a, **k))


exp, sqrt, square, abs = (_un(f) for f in (_np.exp, _np.sqrt, _np.square, _np.abs))  # noqa: A001


def sin(x):
    x = _t(x)
    out = _t(_np.sin(_np.asarray(x)))
    if x._tangent is not None:
        out._tangent = _np.cos(_np.asarray(x))[..., None] * x._tangent
    return out


def cos(x):
    x = _t(x)
    out = _t(_np.cos(_np.asarray(x)))
    if x._tangent is not None:
        out._tangent = -_np.sin(_np.asarray(x))[..., None] * x._tangent
    return out


acos = _un(_np.arccos)


def atan2(y, x):
    return _t(_np.arctan2(_t(y), _t(x)))


def rsqrt(x):
    return _t(_np.float32(1) / _np.sqrt(_t(x)))


def maximum(a, b):
    return _t(_np.maximum(_t(a), _t(b)))


def minimum(a, b):
    return _t(_np.minimum(_t(a), _t(b)))


def multiply(a, b):
    return _t(_t(a) * _t(b))


def equal(a, b):
    return _np.equal(a, b)


def logical_or(a, b):
    return _np.logical_or(a, b)


def logical_and(a, b):
    return _np.logical_and(a, b)


def clip_by_value(x, clip_value_min, clip_value_max):
    x = _t(x)
    return _t(_np.clip(x, _np.asarray(clip_value_min, x.dtype), _np.asarray(clip_value_max, x.dtype)))


def reduce_sum(x, axis=None, keepdims=False):
    return _t(_np.sum(_t(x), axis=axis, keepdims=keepdims, dtype=_t(x).dtype))


def reduce_mean(x, axis=None, keepdims=False):
    return _t(_np.mean(_t(x), axis=axis, keepdims=keepdims, dtype=_t(x).dtype))


def reduce_min(x, axis=None):
    return _t(_np.min(x, axis=axis))


def reduce_max(x, axis=None):
    return _t(_np.max(x, axis=axis))


def cumsum(x, axis=0):
    return _t(_np.cumsum(_t(x), axis=axis, dtype=_t(x).dtype))


def matmul(a, b):
    a, b = _t(a), _t(b)
    out = _t(_np.matmul(_np.asarray(a), _np.asarray(b)))
    if a._tangent is not None:
        assert a.ndim == 2 and b._tangent is None
        out._tangent = _np.einsum('ndk,dm->nmk', a._tangent, _np.asarray(b)).astype(_np.float32)
    return out


def einsum(eq, *ops):
    return _t(_np.einsum(eq, *[_t(o) for o in ops]))


def sort(x, axis=-1):
    return _t(_np.sort(x, axis=axis))


def where(condition, x=None, y=None):
    if x is None:
        return _t(_np.argwhere(condition).astype(_np.int64))
    x, y = _t(x), _t(y)
    return _t(_np.where(condition, x, y).astype(_np.result_type(x, y)))


def boolean_mask(tensor, mask, axis=None):
    assert axis in (None, 0)
    return _t(_np.asarray(tensor)[_np.asarray(mask, _np.bool_)])


def gather(params, indices, axis=0, batch_dims=0):
    params, indices = _np.asarray(params), _np.asarray(indices)
    if batch_dims == 0:
        return _t(_np.take(params, indices, axis=axis))
    # the one form the reference uses: gather along the last axis with all leading axes batched
    assert axis in (-1, params.ndim - 1) and batch_dims == params.ndim - 1, (axis, batch_dims, params.shape)
    flat_p = params.reshape(-1, params.shape[-1])
    flat_i = indices.reshape(flat_p.shape[0], -1)
    out = _np.take_along_axis(flat_p, flat_i, axis=1)
    return _t(out.reshape(indices.shape))


def gather_nd(params, indices):
    indices = _np.asarray(indices)
    return _t(_np.asarray(params)[tuple(indices[..., k] for k in _np.arange(indices.shape[-1]))])


def scatter_nd(indices, updates, shape):  # noqa: A002
    indices, updates = _np.asarray(indices), _np.asarray(updates)
    out = _np.zeros(tuple(int(s) for s in shape), updates.dtype)
    _np.add.at(out, tuple(indices[..., k] for k in _np.arange(indices.shape[-1])), updates)
    return _t(out)


def tensor_scatter_nd_update(tensor, indices, updates):
    out = _np.array(tensor, copy=True)
    indices = _np.asarray(indices)
    out[tuple(indices[..., k] for k in _np.arange(indices.shape[-1]))] = updates
    return _t(out)


def searchsorted(sorted_sequence, values, side='left'):
    seq, val = _np.asarray(sorted_sequence), _np.asarray(values)
    flat_s, flat_v = seq.reshape(-1, seq.shape[-1]), val.reshape(-1, val.shape[-1])
    out = _np.stack([_np.searchsorted(s, v, side=side) for s, v in zip(flat_s, flat_v)])
    return _t(out.reshape(val.shape).astype(_np.int32))


def cond(pred, true_fn, false_fn):
    return true_fn() if pred else false_fn()


class control_dependencies:  # noqa: N801
    def __init__(self, deps):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def custom_gradient(f):
    """Forward value only: the wrapped function returns (value, grad_fn)."""
    def wrapped(*args, **kwargs):
        return f(*args, **kwargs)[0]
    wrapped.__name__ = getattr(f, '__name__', 'custom_gradient')
    return wrapped


class GradientTape:
    """Forward-mode stand-in for the one pattern the reference uses (geometry_from_nerf.py:289-295): watch an [N, 3]
    tensor, push it through positional encoding and Dense/ReLU layers, ask for the batch Jacobian of an [N, M] output.
    Tangents ride on Tensor._tangent through: + - * (by an unwatched factor), tf.sin, tf.cos, tf.concat, tf.matmul,
    tf.nn.relu, tf.identity.  Any other operation drops the tangent and batch_jacobian() then fails loudly."""
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False

    def watch(self, x):
        assert isinstance(x, Tensor) and x.ndim == 2 and x.shape[1] == 3
        x._tangent = _np.broadcast_to(_np.eye(3, dtype=_np.float32), (x.shape[0], 3, 3)).copy()

    def batch_jacobian(self, target, source):
        if getattr(target, '_tangent', None) is None:
            raise RuntimeError('the watched tensor reached the target through an operation this shim cannot '
                               'differentiate')
        return _np.asarray(target._tangent, _np.float32).view(Tensor)


def random_normal_initializer(mean=0., stddev=1.):
    def init(shape, dtype='float32'):  # noqa: A002
        return _t((random._rng.standard_normal(tuple(shape)) * stddev + mean).astype(as_dtype(dtype)))
    return init


# ------------------------------------------------------------------ namespaces
class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def _sigmoid(x):
    x = _t(x)
    return _t((_np.float32(1) / (_np.float32(1) + _np.exp(-x))).astype(x.dtype))


def _cumprod(x, axis=0, exclusive=False):
    x = _t(x)
    out = _np.cumprod(x, axis=axis, dtype=x.dtype)
    if exclusive:
        out = _np.roll(out, 1, axis=axis)
        idx = [slice(None)] * x.ndim
        idx[axis] = 0
        out[tuple(idx)] = 1
    return _t(out)


def _divide_no_nan(a, b):
    a, b = _np.broadcast_arrays(_t(a), _t(b))
    out = _np.zeros(a.shape, _np.result_type(a, b))
    _np.divide(a, b, out=out, where=b != 0)
    return _t(out)


def _l2_normalize(x, axis=None, epsilon=1e-12):
    x = _t(x)
    sq = _np.sum(_np.square(x), axis=axis, keepdims=True, dtype=x.dtype)
    return _t(x * (_np.float32(1) / _np.sqrt(_np.maximum(sq, _np.asarray(epsilon, x.dtype)))))


def _norm(x, ord='euclidean', axis=None, keepdims=False):  # noqa: A002
    x = _t(x)
    return _t(_np.sqrt(_np.sum(_np.square(x), axis=axis, keepdims=keepdims, dtype=x.dtype)))


def _cross(a, b):
    return _t(_np.cross(_t(a), _t(b)))


def _floormod(x, y):
    x = _t(x)
    return _t(_np.mod(x, _np.asarray(y, x.dtype)))


def _relu(x):
    x = _t(x)
    out = _t(_np.maximum(_np.asarray(x), _np.asarray(0, x.dtype)))
    if x._tangent is not None:
        out._tangent = (_np.asarray(x) > 0)[..., None] * x._tangent
    return out


def _softplus(x):
    x = _t(x)
    return _t(_np.logaddexp(x, _np.asarray(0, x.dtype)).astype(x.dtype))


def _pow(x, y):
    x = _t(x)
    return _t(_np.power(x, _np.asarray(y, x.dtype)))


math = _NS(sin=sin, cos=cos, log=_un(_np.log), sigmoid=_sigmoid, cumprod=_cumprod, divide_no_nan=_divide_no_nan,
           floormod=_floormod, l2_normalize=_l2_normalize, minimum=minimum, maximum=maximum, pow=_pow, sqrt=sqrt,
           exp=exp, abs=abs, softplus=_softplus)
linalg = _NS(l2_normalize=_l2_normalize, norm=_norm, cross=_cross)
nn = _NS(relu=_relu, sigmoid=_sigmoid, softplus=_softplus)


class _Random:
    def __init__(self):
        self._rng = _np.random.default_rng(0)

    def set_seed(self, seed):
        self._rng = _np.random.default_rng(seed)

    def uniform(self, shape, minval=0., maxval=1., dtype=float32):  # noqa: A002
        return _t((self._rng.random(tuple(int(s) for s in shape)) * (maxval - minval) + minval).astype(dtype))

    def normal(self, shape, mean=0., stddev=1., dtype=float32):  # noqa: A002
        return _t((self._rng.standard_normal(tuple(int(s) for s in shape)) * stddev + mean).astype(dtype))


random = _Random()


def _assert_greater(x, y, message=None):
    if not _np.all(_np.asarray(x) > y):
        raise AssertionError(message or 'assert_greater failed')


def _check_numerics(x, message):
    if not _np.all(_np.isfinite(x)):
        raise FloatingPointError(message)
    return x


def _debug_assert(condition, data=None, **_):
    if not _np.all(condition):
        raise AssertionError(data)


debugging = _NS(assert_greater=_assert_greater, check_numerics=_check_numerics, Assert=_debug_assert)


def _resize(images, size, method='bilinear', antialias=False):
    images = _t(images)
    if tuple(int(s) for s in size) == tuple(images.shape[-3:-1]):
        return _t(images.astype(_np.float32))
    raise NotImplementedError('tf.image.resize to a different size is outside this shim')


image = _NS(resize=_resize)


class _Restore:
    def expect_partial(self):
        return self


class _Checkpoint:
    """Restoring is a no-op: the golden script assigns weights itself."""
    def __init__(self, **kw):
        self.kw = kw

    def restore(self, path):
        return _Restore()


train = _NS(Checkpoint=_Checkpoint)
data = _NS(experimental=_NS(AUTOTUNE=-1))      # the reference's Dataset.__init__ reads the constant only
string = _np.str_

from . import keras  # noqa: E402,F401
