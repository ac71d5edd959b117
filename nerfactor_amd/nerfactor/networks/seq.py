"""networks.seq.Network (reference: nerfactor/networks/seq.py:24-38): layers applied in order."""
from .base import Network as BaseNetwork


class Network(BaseNetwork):
    def build(self, in_dims):
        d = in_dims
        for layer in self.layers:
            d = layer.build(d)
        return d

    def forward(self, x):
        for layer in self.layers:
            x = layer(x)
        return x
