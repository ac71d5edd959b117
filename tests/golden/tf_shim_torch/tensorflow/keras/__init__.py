"""tf.keras stand-in on torch tensors: Model (with Keras' attribute tracking of layers / variables for
`trainable_variables`), Sequential, the layers and losses of ../../../tf_shim/tensorflow/keras, and
optimizers.Adam(amsgrad=True) as TF 2.2 applies it to dense variables."""
import math as _math

import torch as _torch

from . import layers, losses  # noqa: F401


def _collect(obj, out, seen, only_trainable):
    if id(obj) in seen:
        return
    seen.add(id(obj))
    if isinstance(obj, _torch.Tensor):
        if hasattr(obj, 'trainable') and (obj.trainable or not only_trainable):
            out.append(obj)
    elif isinstance(obj, layers.Layer):
        if obj.trainable or not only_trainable:
            for v in obj.variables():
                _collect(v, out, seen, only_trainable)
            for v in obj.__dict__.values():     # Keras tracks variables assigned as attributes (layers.LatentCode._z)
                if isinstance(v, _torch.Tensor):
                    _collect(v, out, seen, only_trainable)
    elif isinstance(obj, (Model, Sequential)):
        if getattr(obj, 'trainable', True) or not only_trainable:
            for v in obj.__dict__.values():
                _collect(v, out, seen, only_trainable)
    elif isinstance(obj, dict):
        for v in obj.values():
            _collect(v, out, seen, only_trainable)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            _collect(v, out, seen, only_trainable)


class Model:
    def __init__(self, *a, **k):
        self.trainable = True

    def __call__(self, *a, **k):
        return self.call(*a, **k)

    @property
    def trainable_variables(self):
        out = []
        _collect(self, out, set(), True)
        return out


class Sequential:
    def __init__(self, layers=None):  # noqa: A002
        self.layers = list(layers or [])
        self.trainable = True

    def build(self, input_shape):
        dim = int(input_shape[-1])
        for layer in self.layers:
            layer.build((None, dim))
            dim = getattr(layer, 'units', dim)

    def __call__(self, x):
        for layer in self.layers:
            x = layer(x)
        return x


class _ExponentialDecay:
    def __init__(self, initial_learning_rate, decay_steps, decay_rate, staircase=False):
        self.lr0, self.decay_steps, self.decay_rate, self.staircase = initial_learning_rate, decay_steps, decay_rate, staircase

    def __call__(self, step):
        p = step / self.decay_steps
        return self.lr0 * self.decay_rate ** (_math.floor(p) if self.staircase else p)


class _Adam:
    """tf.keras.optimizers.Adam (TF 2.2, optimizer_v2/adam.py, dense `_resource_apply_dense`):
        t = iterations + 1;  lr_t = lr(iterations) * sqrt(1 - beta_2^t) / (1 - beta_1^t)
        m = beta_1 m + (1 - beta_1) g;  v = beta_2 v + (1 - beta_2) g^2
        amsgrad: vhat = max(vhat, v);  var -= lr_t * m / (sqrt(vhat) + epsilon)         (epsilon = 1e-7)
    with the optional `clipnorm` (per gradient tensor) / `clipvalue` of OptimizerV2._compute_gradients."""
    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7, amsgrad=False, clipnorm=None,
                 clipvalue=None, **_):
        self.lr, self.b1, self.b2, self.eps, self.amsgrad = learning_rate, beta_1, beta_2, epsilon, amsgrad
        self.clipnorm, self.clipvalue = clipnorm, clipvalue
        self.iterations = 0
        self.slots = {}

    def apply_gradients(self, grads_and_vars):
        lr = self.lr(self.iterations) if callable(self.lr) else self.lr
        t = self.iterations + 1
        lr_t = lr * _math.sqrt(1 - self.b2 ** t) / (1 - self.b1 ** t)
        with _torch.no_grad():
            for g, var in grads_and_vars:
                if g is None:
                    continue
                if self.clipnorm is not None:
                    g = g * _torch.clamp(self.clipnorm / (g.norm() + 0.), max=1.) if g.norm() > 0 else g
                if self.clipvalue is not None:
                    g = g.clamp(-self.clipvalue, self.clipvalue)
                m, v, vhat = self.slots.setdefault(id(var), [_torch.zeros_like(var) for _ in range(3)])
                m.mul_(self.b1).add_(g, alpha=1 - self.b1)
                v.mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                if self.amsgrad:
                    _torch.maximum(vhat, v, out=vhat)
                    var.sub_(lr_t * m / (vhat.sqrt() + self.eps))
                else:
                    var.sub_(lr_t * m / (v.sqrt() + self.eps))
        self.iterations += 1


class _NS:
    def __init__(self, **kw):
        self.__dict__.update(kw)


optimizers = _NS(Adam=_Adam, schedules=_NS(ExponentialDecay=_ExponentialDecay))
