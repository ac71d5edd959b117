"""networks.embedder.Embedder — sinusoidal positional encoding (reference:
nerfactor/networks/embedder.py:23-47).  Output order: [x, sin(f0 x), cos(f0 x), sin(f1 x), ...].
On the hot path the encoding is computed inside the fused kernels; this class carries the
configuration (`n_freqs`, `out_dims`) and a torch evaluation for off-path callers."""
import torch


class Embedder:
    def __init__(self, incl_input=True, in_dims=3, log2_max_freq=3, n_freqs=4, log_sampling=True,
                 periodic_func=None):
        if periodic_func is None:
            periodic_func = [torch.sin, torch.cos]
        self.incl_input = incl_input
        self.in_dims = in_dims
        self.n_freqs = n_freqs
        self.periodic_func = list(periodic_func)
        if log_sampling:
            self.freq_bands = 2. ** torch.linspace(0., log2_max_freq, n_freqs)
        else:
            self.freq_bands = torch.linspace(2. ** 0., 2. ** log2_max_freq, n_freqs)
        self.out_dims = in_dims * (int(incl_input) + n_freqs * len(self.periodic_func))
        # what the fused kernels implement: input included, bands 2^k, (sin, cos) pairs
        self.fusable = bool(
            incl_input and in_dims == 3 and log_sampling and log2_max_freq == n_freqs - 1 and
            self.periodic_func == [torch.sin, torch.cos])

    def __call__(self, x):
        parts = [x] if self.incl_input else []
        for f in self.freq_bands.tolist():
            for fn in self.periodic_func:
                parts.append(fn(x * f))
        return torch.cat(parts, -1)
