"""networks.layers.LatentCode — per-identity codes optimised jointly with the network (GLO),
reference: nerfactor/networks/layers.py:24-67.  The reference's conv/norm/pool helpers in the same
file are used by no model and are not reproduced."""
import torch

from ..util.math import safe_l2_normalize


def slerp(p0, p1, t):
    """Spherical interpolation of two [1, D] unit vectors (util/geom.py:100-116)."""
    omega = torch.acos(torch.clamp((p0 * p1).sum(), -1., 1.))
    return p0 * torch.sin((1 - t) * omega) / torch.sin(omega) + \
        p1 * torch.sin(t * omega) / torch.sin(omega)


class LatentCode(torch.nn.Module):
    def __init__(self, n_iden, dim, mean=0., std=1., normalize=False):
        super().__init__()
        self._z = torch.nn.Parameter(torch.randn(n_iden, dim) * std + mean)
        self.normalize = normalize

    @property
    def z(self):
        return safe_l2_normalize(self._z, axis=1) if self.normalize else self._z

    @z.setter
    def z(self, value):
        self._z = torch.nn.Parameter(torch.as_tensor(value, dtype=torch.float32))

    def forward(self, ind):
        ind = torch.as_tensor(ind, device=self._z.device).reshape(-1).long()
        return self.z[ind]

    def interp(self, w1, i1, w2, i2):
        z1, z2 = self(i1), self(i2)
        if self.normalize:
            if w1 + w2 != 1.:
                raise ValueError("When latent codes are normalized, use weights that sum to 1")
            return slerp(z1, z2, w2)
        return w1 * z1 + w2 * z2
