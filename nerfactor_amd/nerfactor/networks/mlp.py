"""networks.mlp.Network — Dense stack with skip-concatenation (reference: nerfactor/networks/mlp.py:
24-50).  Parameters keep the Keras layout (`kernel` [in, out], `bias` [out]) so checkpoints and the
libnfx weight packer see exactly the reference's tensors.

`forward` is the plain torch evaluation (used off the hot path and as the GPU fp32 cross-check);
the models never call it per ray — they hand the kernels to the fused HIP kernels instead."""
import math

import torch

from .seq import Network as SeqNetwork

_ACTS = {
    None: None, 'relu': torch.relu, 'sigmoid': torch.sigmoid,
    'softplus': torch.nn.functional.softplus}


class Dense(torch.nn.Module):
    """keras.layers.Dense(units, activation): y = act(x @ kernel + bias); glorot-uniform / zeros."""

    def __init__(self, units, activation=None):
        super().__init__()
        if activation not in _ACTS:
            raise NotImplementedError(activation)
        self.units = units
        self.activation = activation
        self.kernel = None
        self.bias = None

    @property
    def built(self):
        return self.kernel is not None

    @property
    def trainable(self):
        return self.built and self.kernel.requires_grad

    def build(self, in_dims):
        if not self.built:
            lim = math.sqrt(6. / (in_dims + self.units))
            self.kernel = torch.nn.Parameter(
                torch.empty(in_dims, self.units).uniform_(-lim, lim))
            self.bias = torch.nn.Parameter(torch.zeros(self.units))
        return self.units

    def forward(self, x):
        if not self.built:
            self.build(x.shape[-1])
            self.to(x.device)
        y = x @ self.kernel + self.bias
        f = _ACTS[self.activation]
        return y if f is None else f(y)


class Network(SeqNetwork):
    def __init__(self, widths, act=None, skip_at=None):
        super().__init__()
        if act is None:
            act = [None] * len(widths)
        if len(act) != len(widths):
            raise ValueError("If not `None`, `act` must have the same length as `widths`")
        for w, a in zip(widths, act):
            self.layers.append(Dense(w, activation=a))
        self.skip_at = skip_at

    def build(self, in_dims):
        d = in_dims
        for i, layer in enumerate(self.layers):
            d = layer.build(d)
            if self.skip_at is not None and i in self.skip_at:
                d += in_dims
        return d

    def forward(self, x):
        if self.skip_at is None:
            return super().forward(x)
        h = x
        for i, layer in enumerate(self.layers):
            h = layer(h)
            if i in self.skip_at:
                h = torch.cat((h, x), -1)  # activation first, then the original input
        return h

    def kernels_and_biases(self):
        return [l.kernel for l in self.layers], [l.bias for l in self.layers]
