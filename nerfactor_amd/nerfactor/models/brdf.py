"""models.brdf.Model — the MERL BRDF prior: a latent-code-conditioned MLP on Rusinkiewicz
coordinates (reference: nerfactor/models/brdf.py:34-136).  NeRFactor evaluates its frozen
`brdf_mlp`/`brdf_out` per (surface point, light) inside nfx_brdf_spec_fwd; this class owns the
weights, the latent codes and the config surface."""
from os.path import basename
import glob

import torch

from ..networks import mlp
from ..networks.embedder import Embedder
from ..networks.layers import LatentCode
from .base import Model as BaseModel


class Model(BaseModel):
    def __init__(self, config, debug=False):
        super().__init__(config, debug=debug)
        cfg = self.config
        self.mlp_chunk = cfg.getint('DEFAULT', 'mlp_chunk')
        self.z_dim = cfg.getint('DEFAULT', 'z_dim')
        self.embedder = self._init_embedder()
        self.net = self._init_net()
        data_dir = cfg.get('DEFAULT', 'data_root', fallback='')
        train_npz = sorted(glob.glob(data_dir.rstrip('/') + '/train_*.npz')) if data_dir else []
        self.brdf_names = [basename(x)[len('train_'):-len('.npz')] for x in train_npz]
        n_brdfs = max(1, len(self.brdf_names))
        self.latent_code = LatentCode(
            n_brdfs, self.z_dim, mean=cfg.getfloat('DEFAULT', 'z_gauss_mean'),
            std=cfg.getfloat('DEFAULT', 'z_gauss_std'),
            normalize=cfg.getboolean('DEFAULT', 'normalize_z'))
        self.register_trainable()

    def _init_net(self):
        cfg = self.config
        width = cfg.getint('DEFAULT', 'mlp_width')
        depth = cfg.getint('DEFAULT', 'mlp_depth')
        skip_at = cfg.getint('DEFAULT', 'mlp_skip_at')
        if (width, depth, skip_at) != (128, 4, 2):
            raise NotImplementedError("libnfx implements mlp_width=128, mlp_depth=4, mlp_skip_at=2")
        body = mlp.Network([width] * depth, act=['relu'] * depth, skip_at=[skip_at])
        body.build(self.z_dim + self.embedder['rusink'].out_dims)
        head = mlp.Network([1], act=['softplus'])  # reflectance > 0
        head.build(width)
        return {'brdf_mlp': body, 'brdf_out': head}

    def _init_embedder(self):
        cfg = self.config
        if not cfg.getboolean('DEFAULT', 'pos_enc'):
            raise NotImplementedError("pos_enc=False is not supported by the fused kernels")
        n_freqs = cfg.getint('DEFAULT', 'n_freqs')
        if n_freqs != 2:
            raise NotImplementedError("libnfx implements the shipped Rusinkiewicz encoder (n_freqs=2)")
        return {'rusink': Embedder(incl_input=True, in_dims=3, log2_max_freq=n_freqs - 1,
                                   n_freqs=n_freqs)}

    def _eval_brdf_at(self, z, rusink):
        """Explicit-row evaluation (z [M, z_dim], rusink [M, 3]) -> (brdf, brdf_reci) [M, 1]; plain
        torch (the MERL prior is trained once, off the per-ray hot path — SURVEY.md §8f-4)."""
        body, head = self.net['brdf_mlp'], self.net['brdf_out']
        emb = self.embedder['rusink']
        brdf = head(body(torch.cat((z, emb(rusink)), 1)))
        reci = torch.cat((rusink[:, :1] + torch.pi, rusink[:, 1:]), 1)  # reciprocity: phi_d + pi
        brdf_reci = head(body(torch.cat((z, emb(reci)), 1)))
        return brdf, brdf_reci
