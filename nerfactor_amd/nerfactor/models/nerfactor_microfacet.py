"""models.nerfactor_microfacet.Model — NeRFactor with the analytic GGX microfacet BRDF instead of
the learned prior; the BRDF latent is one roughness scalar squashed to [0, 1] (reference:
nerfactor/models/nerfactor_microfacet.py:34-132).  The BRDF is evaluated inside the shading
kernels (nerfactor_amd/csrc/geom.hpp)."""
import torch

from nerfactor_amd import autograd as nfx_grad
from nerfactor_amd.brdf.microfacet.microfacet import Microfacet

from .. import config as default_configs
from ..util import config as configutil
from .nerfactor import Model as NeRFactorModel
from .shape import Model as ShapeModel


class Model(NeRFactorModel):
    def __init__(self, config, debug=False):
        self.pred_brdf = config.getboolean('DEFAULT', 'pred_brdf')
        self.z_dim = 1  # scalar roughness
        self.normalize_brdf_z = False
        self.shape_mode = config.get('DEFAULT', 'shape_mode')
        self.shape_model_ckpt = config.get('DEFAULT', 'shape_model_ckpt', fallback='none')
        self.config_shape = None
        if self.shape_mode not in ('nerf', 'scratch'):
            self.config_shape = self._load_sub_config(self.shape_model_ckpt, 'shape')
        # grandparent construction: no BRDF prior, no Rusinkiewicz embedder
        ShapeModel.__init__(self, config, debug=debug)
        self.albedo_smooth_weight = config.getfloat('DEFAULT', 'albedo_smooth_weight')
        self.brdf_smooth_weight = config.getfloat('DEFAULT', 'brdf_smooth_weight')
        self._init_lighting()

    def _init_embedder(self):
        return ShapeModel._init_embedder(self)

    @staticmethod
    def _brdf_z_act():
        return 'sigmoid'

    def _brdf_terms(self, xyz, cam, normal, brdf_prop):
        return {'rough': brdf_prop, 'f0': self.config.getfloat('DEFAULT', 'fresnel_f0')}

    def _render_train(self, xyz, cam, normal, albedo, brdf_prop, light_vis, light, to_srgb):
        return nfx_grad.ShadeMicrofacet.apply(
            xyz, cam, self.lxyz.reshape(-1, 3), self.lareas, self.config.getfloat('DEFAULT', 'fresnel_f0'),
            to_srgb, normal, albedo, brdf_prop, light_vis, light)

    def _eval_brdf_at(self, pts2l, pts2c, normal, albedo, brdf_prop, xyz=None, cam=None):
        """Explicit [N, L, 3] tensor via the torch Microfacet (off the hot path)."""
        microfacet = Microfacet(f0=self.config.getfloat('DEFAULT', 'fresnel_f0'))
        brdf = microfacet(pts2l, pts2c, normal, albedo=albedo, rough=brdf_prop)
        if not torch.isfinite(brdf).all():
            raise FloatingPointError("BRDF")
        return brdf
