"""tf.keras stand-in: Model / Sequential containers, the Dense / Activation / Layer classes and the two losses the
reference's model code uses.  Weights are plain float32 arrays (`layer.kernel`, `layer.bias`)."""
from . import layers, losses  # noqa: F401


class Model:
    def __init__(self, *a, **k):
        self.trainable = True

    def __call__(self, *a, **k):
        return self.call(*a, **k)


class Sequential:
    def __init__(self, layers=None):  # noqa: A002
        self.layers = list(layers or [])

    def build(self, input_shape):
        dim = int(input_shape[-1])
        for layer in self.layers:
            layer.build((None, dim))
            dim = getattr(layer, 'units', dim)

    def __call__(self, x):
        for layer in self.layers:
            x = layer(x)
        return x
