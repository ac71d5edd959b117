import numpy as _np

_ACT = {}


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

    def __call__(self, *a, **k):
        if not self.built:
            first = a[0]
            self.build(getattr(first, 'shape', None))
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
    """y = activation(x @ kernel + bias); glorot-uniform kernel, zero bias (the Keras defaults)."""
    def __init__(self, units, activation=None, use_bias=True):
        super().__init__()
        self.units = int(units)
        if isinstance(activation, (str, type(None))):
            activation = _activation(activation)
        self.activation = activation
        self.kernel = None
        self.bias = None

    def build(self, input_shape):
        import tensorflow as tf
        fan_in = int(input_shape[-1])
        limit = _np.sqrt(6. / (fan_in + self.units))
        self.kernel = tf.random.uniform((fan_in, self.units), -limit, limit)
        self.bias = tf.zeros((self.units,))
        self.built = True

    def set_weights(self, weights):
        import tensorflow as tf
        kernel, bias = weights
        assert kernel.shape[1] == self.units and bias.shape == (self.units,), (kernel.shape, bias.shape, self.units)
        self.kernel = tf.convert_to_tensor(_np.asarray(kernel, _np.float32))
        self.bias = tf.convert_to_tensor(_np.asarray(bias, _np.float32))
        self.built = True

    def get_weights(self):
        return [_np.asarray(self.kernel), _np.asarray(self.bias)]

    def call(self, x):
        import tensorflow as tf
        assert x.shape[-1] == self.kernel.shape[0], (x.shape, self.kernel.shape)
        assert x.dtype == _np.float32, x.dtype
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
