import numpy as _np
import torch as _torch


def _activation(name):
    import tensorflow as tf
    table = {'relu': tf.nn.relu, 'sigmoid': tf.math.sigmoid, 'softplus': tf.math.softplus,
             'linear': tf.identity, None: tf.identity}
    return table[name]


class Layer:
    def __init__(self, *a, **k):
        self.trainable = True
        self.built = False

    def build(self, input_shape):
        self.built = True

    def variables(self):
        return []

    def __call__(self, *a, **k):
        if not self.built:
            self.build(getattr(a[0], 'shape', None))
        return self.call(*a, **k)


class Activation(Layer):
    def __init__(self, activation):
        super().__init__()
        self.fn = _activation(activation) if isinstance(activation, (str, type(None))) else activation
        self.built = True

    def call(self, x):
        return self.fn(x)


class Lambda(Layer):
    def __init__(self, function):
        super().__init__()
        self.fn = function
        self.built = True

    def call(self, x):
        return self.fn(x)


class Dense(Layer):
    """y = activation(x @ kernel + bias); glorot-uniform kernel, zero bias (the Keras defaults); kernel and bias are
    trainable variables (leaf tensors with requires_grad)."""
    def __init__(self, units, activation=None, use_bias=True):
        super().__init__()
        self.units = int(units)
        if isinstance(activation, (str, type(None))):
            activation = _activation(activation)
        self.activation = activation
        self.kernel = None
        self.bias = None

    def _var(self, array):
        import tensorflow as tf
        return tf.Variable(_np.asarray(array, _np.float32), trainable=True)

    def build(self, input_shape):
        import tensorflow as tf
        fan_in = int(input_shape[-1])
        limit = _np.sqrt(6. / (fan_in + self.units))
        self.kernel = self._var(tf.random.uniform((fan_in, self.units), -limit, limit).numpy())
        self.bias = self._var(_np.zeros(self.units, _np.float32))
        self.built = True

    def set_weights(self, weights):
        kernel, bias = weights
        assert kernel.shape[1] == self.units and bias.shape == (self.units,), (kernel.shape, bias.shape, self.units)
        self.kernel, self.bias = self._var(kernel), self._var(bias)
        self.built = True

    def get_weights(self):
        return [self.kernel.detach().numpy(), self.bias.detach().numpy()]

    def variables(self):
        return [self.kernel, self.bias] if self.built else []

    def call(self, x):
        import tensorflow as tf
        assert x.shape[-1] == self.kernel.shape[0], (x.shape, self.kernel.shape)
        assert x.dtype == _torch.float32, x.dtype
        return self.activation(tf.matmul(x, self.kernel) + self.bias)


def _unsupported(name):
    class _U(Layer):
        def __init__(self, *a, **k):
            raise NotImplementedError('tf.keras.layers.%s is outside this shim' % name)
    _U.__name__ = name
    return _U


for _n in ('Conv2D', 'Conv2DTranspose', 'UpSampling2D', 'MaxPooling2D', 'AveragePooling2D', 'BatchNormalization',
           'LayerNormalization', 'LeakyReLU', 'ELU', 'ReLU'):
    globals()[_n] = _unsupported(_n)
