"""models.nerfactor.Model — the NeRFactor factorisation forward (reference:
nerfactor/models/nerfactor.py:33-541): surface points -> normals, light visibility, albedo, BRDF
latent -> learned-BRDF evaluation -> rendering integral (+ relighting) -> losses.

Per-ray work runs in libnfx:
    normals / albedo / z      nfx_mlp128_xyz_fwd
    light visibility          nfx_lvis_fwd            (N x L rows, posenc(xyz) folded per point)
    learned BRDF (specular)   nfx_brdf_spec_fwd       (world->local, Rusinkiewicz, frozen MLP)
    render + light probes     nfx_shade_fwd           (all probes in one pass over the sphere)
    OLAT relighting           nfx_shade_olat_fwd
Masking / compaction / scatter of the alpha > 0 rays is torch indexing (plumbing).
"""
from collections import OrderedDict
import glob
from os.path import basename, join

import numpy as np
import torch

from nerfactor_amd import _capi, autograd as nfx_grad, ops

from .. import config as default_configs
from ..datasets.nerf_shape import known_all_foreground
from ..util import config as configutil, img as imgutil, light as lightutil, math as mathutil
from .brdf import Model as BRDFModel
from .shape import Model as ShapeModel, _mae, _mse


class Model(ShapeModel):
    def __init__(self, config, debug=False):
        # --- BRDF prior (frozen): its own config sits next to its checkpoint
        brdf_ckpt = config.get('DEFAULT', 'brdf_model_ckpt', fallback='none')
        self.config_brdf = self._load_sub_config(brdf_ckpt, 'brdf')
        self.pred_brdf = config.getboolean('DEFAULT', 'pred_brdf')
        self.z_dim = self.config_brdf.getint('DEFAULT', 'z_dim')
        self.normalize_brdf_z = self.config_brdf.getboolean('DEFAULT', 'normalize_z')
        # --- shape
        self.shape_mode = config.get('DEFAULT', 'shape_mode')
        self.shape_model_ckpt = config.get('DEFAULT', 'shape_model_ckpt', fallback='none')
        self.config_shape = None
        if self.shape_mode not in ('nerf', 'scratch'):
            self.config_shape = self._load_sub_config(self.shape_model_ckpt, 'shape')
        self._brdf_ckpt = brdf_ckpt
        super().__init__(config, debug=debug)
        self.albedo_smooth_weight = config.getfloat('DEFAULT', 'albedo_smooth_weight')
        self.brdf_smooth_weight = config.getfloat('DEFAULT', 'brdf_smooth_weight')
        self._init_brdf_model()
        self._init_lighting()

    # ------------------------------------------------------------------ construction helpers
    @staticmethod
    def _load_sub_config(ckpt, default_name):
        if configutil.ckpt_available(ckpt):
            return configutil.read_config(configutil.get_config_ini(ckpt))
        return default_configs.make_config(default_name)

    def _init_brdf_model(self):
        self.brdf_model = BRDFModel(self.config_brdf)
        if not self.brdf_model.tuned:
            # A prior of another shape than the shipped one (the reference builds whatever brdf.ini says,
            # nerfactor.py:45-60) is evaluated on explicit rows: nfx_brdf_rows_geom_fwd / _bwd + the runtime-shaped MLP
            # kernels (`_brdf_spec_rows`) instead of the fused shading kernels.  What that path needs:
            emb = self.brdf_model.embedder['rusink']
            zd = self.config_brdf.getint('DEFAULT', 'z_dim')
            if not (getattr(emb, 'fusable', False) or (emb.incl_input and emb.n_freqs == 0)) or emb.n_freqs > 8 or zd > 8:
                raise NotImplementedError(
                    "a non-shipped BRDF prior inside NeRFactor needs z_dim <= 8 and a standard Rusinkiewicz embedder "
                    "(input included, bands 2^k with k < n_freqs <= 8, sin / cos): nfx_brdf_rows_geom_fwd builds its rows")
        if configutil.ckpt_available(self._brdf_ckpt):
            configutil.restore_model(self.brdf_model, self._brdf_ckpt)
        for p in self.brdf_model.parameters():
            p.requires_grad_(False)  # the prior stays frozen (nerfactor.py:60)

    def _init_lighting(self):
        cfg = self.config
        light_h = cfg.getint('DEFAULT', 'light_h')
        self.light_res = (light_h, 2 * light_h)
        maxv = cfg.getfloat('DEFAULT', 'light_init_max')
        self._light = torch.nn.Parameter(torch.rand(self.light_res + (3,)) * maxv)
        olat_inten = cfg.getfloat('DEFAULT', 'olat_inten', fallback=200)
        ambi_inten = cfg.getfloat('DEFAULT', 'ambient_inten', fallback=0)
        self.olat_inten = olat_inten
        self.olat_ambient = ambi_inten if self.white_bg else 0.
        # (1) OLAT conditions: names only — the kernel never materialises the one-hot maps
        n_i = 2 if self.debug else self.light_res[0]
        n_j = 2 if self.debug else self.light_res[1]
        self.novel_olat = OrderedDict(
            ('%04d-%04d' % (i, j), (i, j)) for i in range(n_i) for j in range(n_j))
        # (2) light probes: the .hdr / .exr files of test_envmap_dir (nerfactor.py:88-93), resized to light_res with the
        # reference's antialiased bilinear filter (:169-179); .npy arrays are accepted as an extension
        self.novel_probes = OrderedDict()
        envmap_dir = cfg.get('DEFAULT', 'test_envmap_dir', fallback='')
        paths = []
        for ext in ('hdr', 'exr', 'npy'):
            paths += glob.glob(join(envmap_dir, '*.' + ext)) if envmap_dir else []
        for path in sorted(paths):
            self.add_probe(basename(path)[:-len('.hdr')], self._load_light(path))
        self.embed_light_h = cfg.getint('DEFAULT', 'embed_light_h', fallback=32)

    def _load_light(self, path):
        """[light_h, 2 light_h, 3] float32 array of one probe file (nerfactor.py:169-179)."""
        return lightutil.resize_antialias(lightutil.read_probe(path), new_h=self.light_res[0])

    def add_probe(self, name, envmap):
        """Registers a novel light probe; maps of another resolution are resized like the files of test_envmap_dir."""
        env = np.asarray(envmap, np.float32)
        if env.ndim != 3 or env.shape[2] != 3:
            raise ValueError("probe %s has shape %s, expected [h, w, 3]" % (name, tuple(env.shape)))
        if tuple(env.shape[:2]) != self.light_res:
            env = lightutil.resize_antialias(env, new_h=self.light_res[0])
            if tuple(env.shape[:2]) != self.light_res:
                raise ValueError("probe %s is not a 2:1 latitude-longitude map" % name)
        self.novel_probes[name] = torch.from_numpy(np.ascontiguousarray(env))

    def olat_envmap(self, name):
        """The [h, w, 3] environment map of one OLAT condition (what the reference stores)."""
        i, j = self.novel_olat[name]
        env = torch.full(self.light_res + (3,), float(self.olat_ambient))
        env[i, j, :] += self.olat_inten
        return env

    def _init_embedder(self):
        from ..networks.embedder import Embedder
        embedder = super()._init_embedder()
        n = self.config_brdf.getint('DEFAULT', 'n_freqs')  # the level the BRDF MLP was trained with
        embedder['rusink'] = Embedder(incl_input=True, in_dims=3, log2_max_freq=n - 1, n_freqs=n)
        return embedder

    def _init_net(self):
        dx = self.embedder['xyz'].out_dims
        net = {}
        net['albedo_mlp'], net['albedo_out'] = self._mlp128(dx, 3, 'sigmoid')
        if self.pred_brdf:
            net['brdf_z_mlp'], net['brdf_z_out'] = self._mlp128(dx, self.z_dim, self._brdf_z_act())
        if self.shape_mode == 'scratch':
            net.update(super()._init_net())
        elif self.shape_mode in ('frozen', 'finetune'):
            shape_model = ShapeModel(self.config_shape)
            if configutil.ckpt_available(self.shape_model_ckpt):
                configutil.restore_model(shape_model, self.shape_model_ckpt)
            for p in shape_model.parameters():
                p.requires_grad_(self.shape_mode == 'finetune')
            for k in ('normal_mlp', 'normal_out', 'lvis_mlp', 'lvis_out'):
                net[k] = shape_model.net[k]
        elif self.shape_mode != 'nerf':
            raise ValueError(self.shape_mode)
        return net

    @staticmethod
    def _brdf_z_act():
        return None  # linear latent code; the microfacet variant squashes roughness to [0, 1]

    # ------------------------------------------------------------------ lighting
    @property
    def light(self):
        return torch.clamp(self._light, min=0.)  # no negative light

    # ------------------------------------------------------------------ forward
    def call(self, batch, mode='train', relight_olat=False, relight_probes=False,
             albedo_scales=None, albedo_override=None, brdf_z_override=None, xyz_noise=None):
        """`xyz_noise` (extra to the reference signature): the [n_masked, 3] jitter to use instead of
        drawing tf.random.normal-style noise internally — lets a test feed an oracle the same noise."""
        self._validate_mode(mode)
        if mode != 'train' and torch.is_grad_enabled():
            with torch.no_grad():  # vali / test never differentiate (trainvali.py:301-308, test.py:189)
                return self.call(batch, mode, relight_olat, relight_probes, albedo_scales, albedo_override,
                                 brdf_z_override, xyz_noise)
        xyz_jitter_std = self.config.getfloat('DEFAULT', 'xyz_jitter_std')
        id_, hw, rayo, _, rgb, alpha, xyz, normal, lvis = batch
        n_all = alpha.shape[0]
        # training batches are foreground rays only (datasets/nerf_shape.py:102-107) and say so: no compaction, and
        # above all no torch.nonzero, whose row count the host can only read after the whole previous step has drained
        if known_all_foreground(alpha):
            idx, all_fg = None, True
        else:
            idx = torch.nonzero(alpha[:, 0] > 0)[:, 0]  # 100 % background rays are dropped
            all_fg = idx.numel() == n_all
        rgb_all, normal_all, lvis_all = rgb, normal, lvis
        if not all_fg:                  # no gather here and no zero-filled scatter at the end (~40 tiny launches)
            rayo, rgb, xyz, normal = (t[idx].contiguous() for t in (rayo, rgb, xyz, normal))
            # the [n, 512] ground-truth visibility is only gathered where it is an INPUT (shape_mode = nerf); as an
            # output (gt['lvis'] = scatter of the gathered rows) it is one masked copy, not gather + fill + scatter
            lvis = lvis[idx].contiguous() if self.shape_mode == 'nerf' else None
        # The reference also evaluates the jittered copies in vali/test mode (nerfactor.py:198-232)
        # although only the training loss reads them; they are skipped here outside training.
        jitter = xyz_jitter_std > 0 and mode == 'train'
        if jitter:
            xyz_j = xyz + (torch.randn_like(xyz) * xyz_jitter_std if xyz_noise is None else xyz_noise)
        else:
            xyz_j = None
        # The clean and the jittered evaluation of a head share ONE kernel launch per direction (forward, backward,
        # weight gradients): the two point sets are concatenated and the result is split again.
        # (the concatenations are built once for all four heads; torch.split's backward is one cat, two slices' is
        # zeros + copy twice and an add)
        xyz_both = torch.cat((xyz, xyz_j)) if jitter else None
        doubled = {}

        def both(fn, **kw):
            if not jitter:
                return fn(xyz, **kw), None
            for k, v in kw.items():
                if k not in doubled:
                    doubled[k] = torch.cat((v, v))
            out = fn(xyz_both, **{k: doubled[k] for k in kw})
            return torch.split(out, [xyz.shape[0], xyz.shape[0]])      # (sizes, not a chunk length: a 0-row batch still gives two pieces)
        # ------ normals
        if self.shape_mode == 'nerf':
            normal_pred, normal_jitter = normal, None
        else:
            normal_pred, normal_jitter = both(self._pred_normal_at)
        normal_pred = self._normalize(normal_pred)
        if normal_jitter is not None:
            normal_jitter = self._normalize(normal_jitter)
        # ------ light visibility (the jittered points keep the light directions of the clean ones, shape.py:160-163)
        row_of = []   # [n_all] int32: compact row of every ray, -1 for the background (built on first use)

        def rows():
            if not row_of:
                r = torch.full((n_all,), -1, dtype=torch.int32, device=xyz.device)
                r[idx] = torch.arange(idx.numel(), dtype=torch.int32, device=xyz.device)
                row_of.append(r)
            return row_of[0]
        lvis_row = lvis_full = None
        if self.shape_mode == 'nerf':
            lvis_pred, lvis_jitter = torch.clamp(lvis, 1e-8, 1.), None
        elif not all_fg and not jitter and xyz.shape[0] > 0 and self._lvis_rows_ok():
            # a render with background rays (round 6): the kernel stores every visibility at its FINAL row of pred['lvis'] and
            # raises the NaN flag itself — no compact [n, 512] tensor, no scatter pass over it (0.7 ms per 800 x 800 view), no
            # check_numerics pass (0.13 ms); the shading kernels read the rows through the same index
            lvis_row = idx.to(torch.int32)
            lvis_full = torch.empty((n_all, self.lxyz.reshape(-1, 3).shape[0]), dtype=torch.float32, device=xyz.device)
            ops.zero_rows(lvis_full, rows())
            lvis_pred, lvis_jitter = self._pred_lvis_rows(xyz, lvis_full, lvis_row, dir_pts=xyz), None
        else:
            lvis_pred, lvis_jitter = both(self._pred_lvis_at, dir_pts=xyz)
        # ------ albedo
        albedo, albedo_jitter = both(self._pred_albedo_at)
        if albedo_scales is not None:
            albedo = torch.as_tensor(albedo_scales, device=albedo.device).reshape(1, 3) * albedo
        if albedo_override is not None:
            ao = torch.as_tensor(albedo_override, dtype=torch.float32, device=albedo.device)
            albedo = ao[None, :].expand(albedo.shape[0], -1).contiguous() if ao.dim() == 1 else (ao if all_fg else ao[idx])
        # ------ BRDF latent
        if not self.pred_brdf:
            raise NotImplementedError("pred_brdf=False: the reference calls an undefined "
                                      "_get_default_brdf_at (nerfactor.py:256)")
        brdf_prop, brdf_prop_jitter = both(self._pred_brdf_at)
        if self.normalize_brdf_z:
            brdf_prop = self._normalize(brdf_prop)
            if brdf_prop_jitter is not None:
                brdf_prop_jitter = self._normalize(brdf_prop_jitter)
        if brdf_z_override is not None:
            zo = torch.as_tensor(brdf_z_override, dtype=torch.float32, device=xyz.device)
            brdf_prop = zo.reshape(1, self.z_dim).expand(brdf_prop.shape[0], -1).contiguous()
        # ------ rendering equation
        rgb_pred, rgb_olat, rgb_probes = self._render(
            xyz, rayo, normal_pred, albedo, brdf_prop, lvis_pred, relight_olat=relight_olat,
            relight_probes=relight_probes, lvis_row=lvis_row, row_of=rows() if lvis_row is not None else None)

        def full(v):  # zero-filled scatter back to all rays (tf.scatter_nd)
            if v is None or all_fg or v.shape[0] == n_all:   # (n_all rows while some rays are background: the kernel stored
                return v                                     #  the foreground rows at their final place — lvis, rgb_olat)
            if v.is_cuda and v.dtype == torch.float32 and not (torch.is_grad_enabled() and v.requires_grad):
                # one pass that writes every output row once (nfx_scatter_rows) instead of zeros + index_put_
                return ops.scatter_rows(v.contiguous(), rows(), n_all)
            out = torch.zeros((n_all,) + tuple(v.shape[1:]), dtype=v.dtype, device=v.device)
            out[idx] = v
            return out

        pred = {'rgb': full(rgb_pred), 'normal': full(normal_pred), 'lvis': full(lvis_pred),
                'albedo': full(albedo), 'brdf': full(brdf_prop)}
        if rgb_olat is not None:
            pred['rgb_olat'] = full(rgb_olat)
        if rgb_probes is not None:
            pred['rgb_probes'] = full(rgb_probes)
        if all_fg:
            gt = {'rgb': rgb, 'normal': normal, 'lvis': lvis, 'alpha': alpha}
        else:   # tf.scatter_nd of the foreground rows = the inputs with the background rows zeroed
            fg = alpha[:, :1] > 0
            gt = {'rgb': torch.where(fg, rgb_all, torch.zeros_like(rgb_all)),
                  'normal': torch.where(fg, normal_all, torch.zeros_like(normal_all)),
                  'lvis': torch.where(fg, lvis_all, torch.zeros_like(lvis_all[:1, :1])), 'alpha': alpha}
        loss_kwargs = {
            'mode': mode, 'normal_jitter': full(normal_jitter), 'lvis_jitter': full(lvis_jitter),
            'brdf_prop_jitter': full(brdf_prop_jitter), 'albedo_jitter': full(albedo_jitter)}
        to_vis = {'id': id_, 'hw': hw}
        for k, v in pred.items():
            to_vis['pred_' + k] = v
        for k, v in gt.items():
            to_vis['gt_' + k] = v
        return pred, gt, loss_kwargs, to_vis

    @staticmethod
    def _normalize(v):
        """safe_l2_normalize(v, axis=1) (csrc/regularizers.hip: one launch, one more for the pull-back)."""
        return nfx_grad.l2_normalize(v, 1e-6)

    # ------------------------------------------------------------------ heads
    def _pred_albedo_at(self, pts):
        scale = self.config.getfloat('DEFAULT', 'albedo_slope', fallback=0.7)
        bias = self.config.getfloat('DEFAULT', 'albedo_bias', fallback=0.1)
        albedo = self._mlp128_xyz(pts, 'albedo_mlp', 'albedo_out', 3, out_act='sigmoid', post_scale=scale,
                                  post_bias=bias)
        return self.check_numerics(albedo, "Albedo")

    def _pred_brdf_at(self, pts):
        return self._mlp128_xyz(pts, 'brdf_z_mlp', 'brdf_z_out', self.z_dim, out_act=self._brdf_z_act())

    # ------------------------------------------------------------------ BRDF + rendering
    def _brdf_terms(self, xyz, cam, normal, brdf_prop):
        """kwargs for the shading kernels describing the BRDF: learned specular term [N, L]."""
        if not self.brdf_model.tuned:      # a prior of a non-shipped shape: explicit rows + the runtime-shaped kernels
            spec = self._brdf_spec_rows(xyz, cam, normal, brdf_prop)
            return {'spec': spec, 'spec_scale': self.config.getfloat('DEFAULT', 'learned_brdf_scale')}
        blob = self._blob128('brdf_mlp', 'brdf_out', _capi.IN_Z_RUSINK, 1, z_dim=self.z_dim,
                             nets=self.brdf_model.net)
        spec = ops.brdf_spec_fwd(xyz, cam, normal, brdf_prop, self.lxyz.reshape(-1, 3), blob,
                                 prec=self.precision)
        return {'spec': spec, 'spec_scale': self.config.getfloat('DEFAULT', 'learned_brdf_scale')}

    def _eval_brdf_at(self, pts2l, pts2c, normal, albedo, brdf_prop, xyz=None, cam=None):
        """[N, L, 3] BRDF values from explicit directions — the reference signature (nerfactor.py:413-461).  The renderer
        does not use this (it hands the terms of `_brdf_terms` to the fused kernels, which recompute the directions
        from points); given `xyz` and `cam` the fused kernel is used here too, otherwise the frozen prior is evaluated
        on the given directions: local frames, Rusinkiewicz angles (nfx_dir2rusink), front-lit rows through the MLP."""
        scale = self.config.getfloat('DEFAULT', 'learned_brdf_scale')
        if xyz is not None and cam is not None:
            t = self._brdf_terms(xyz, cam, normal, brdf_prop)
            return albedo[:, None, :] / np.pi + t['spec'][:, :, None] * t['spec_scale']
        from ..util import geom as geomutil
        n, nl = pts2l.shape[:2]
        rot = geomutil.gen_world2local(normal)
        vdir = torch.einsum('jkl,jl->jk', rot, pts2c)
        ldir = torch.einsum('jkl,jnl->jnk', rot, pts2l).reshape(-1, 3)
        vrep = vdir[:, None, :].expand(n, nl, 3).reshape(-1, 3)
        front = ldir[:, 2] > 0
        spec = torch.zeros(n * nl, dtype=albedo.dtype, device=albedo.device)
        if bool(front.any()):
            rusink = geomutil.dir2rusink(ldir[front].contiguous(), vrep[front].contiguous())
            z = brdf_prop[:, None, :].expand(n, nl, brdf_prop.shape[1]).reshape(-1, brdf_prop.shape[1])[front]
            with torch.no_grad():
                spec[front] = self.brdf_model._eval_brdf_at(z, rusink)[0][:, 0]
        return albedo[:, None, :] / np.pi + spec.reshape(n, nl, 1).expand(n, nl, 3) * scale

    def _render(self, xyz, cam, normal, albedo, brdf_prop, light_vis, relight_olat=False,
                relight_probes=False, white_light_override=False, white_lvis_override=False, lvis_row=None, row_of=None):
        """`lvis_row` (round 6, inference only): light_vis is a full-size [n_all, L] buffer and row lvis_row[i] of it belongs
        to point i (ops.lvis_fwd(out=, out_row=)); `row_of` [n_all] int32 is its inverse (-1: background).  The OLAT renders
        — [n, 512, 3], the largest tensor of the model — are then stored at their final rows as well and come back full-size."""
        to_srgb = self.config.getboolean('DEFAULT', 'linear2srgb')
        light = torch.ones_like(self.light) if white_light_override else self.light
        if white_lvis_override:
            light_vis = torch.ones_like(light_vis)
        if torch.is_grad_enabled() and any(
                t.requires_grad for t in (normal, albedo, brdf_prop, light_vis, light)):
            if relight_olat or relight_probes:
                raise NotImplementedError("relighting is inference-only; run it under torch.no_grad()")
            return self._render_train(xyz, cam, normal, albedo, brdf_prop, light_vis, light, to_srgb), None, None
        terms = self._brdf_terms(xyz, cam, normal, brdf_prop)
        lights = [light.reshape(-1, 3)]
        if relight_probes:
            lights += [p.to(light.device).reshape(-1, 3) for p in self.novel_probes.values()]
        common = (xyz, cam, normal, albedo, light_vis, self.lxyz.reshape(-1, 3), self.lareas)
        terms = dict(terms, lvis_row=lvis_row)
        out = ops.shade_fwd(*common, torch.stack(lights).detach(), linear2srgb=to_srgb, **terms)
        rgb = out[:, 0]
        rgb_probes = out[:, 1:] if relight_probes else None
        rgb_olat = None
        if relight_olat:
            n_l = self.lxyz.reshape(-1, 3).shape[0]
            if lvis_row is not None and row_of is not None and len(self.novel_olat) == n_l:
                # stored at the final rows by the kernel, NaN flag from its epilogue (before the clip, where tf.clip_by_value
                # would still show it): no [n, 512, 3] scatter (6.3 GB of traffic per 800 x 800 view), no check_numerics pass
                rgb_olat = torch.empty((row_of.shape[0], n_l, 3), dtype=torch.float32, device=xyz.device)
                ops.zero_rows(rgb_olat.view(row_of.shape[0], n_l * 3), row_of)
                flag = torch.zeros(1, dtype=torch.int32, device=xyz.device)
                ops.shade_olat_fwd(*common, self.olat_inten, self.olat_ambient, linear2srgb=to_srgb, out=rgb_olat,
                                   out_row=lvis_row, nan_flag=flag, **terms)
                self.check_flag(flag == 0, "OLAT Renders")
            else:
                rgb_olat = ops.shade_olat_fwd(*common, self.olat_inten, self.olat_ambient,
                                              linear2srgb=to_srgb, **terms)
                if len(self.novel_olat) != rgb_olat.shape[1]:  # debug mode keeps the 2x2 corner only
                    keep = [i * self.light_res[1] + j for (i, j) in self.novel_olat.values()]
                    rgb_olat = rgb_olat[:, keep]
                self.check_numerics(rgb_olat, "OLAT Renders")
        if rgb_probes is not None:
            self.check_numerics(rgb_probes, "Light Probe Renders")
        return rgb, rgb_olat, rgb_probes

    def _render_train(self, xyz, cam, normal, albedo, brdf_prop, light_vis, light, to_srgb):
        """Differentiable render under the trained light: frozen learned BRDF (gradients reach the
        latent z and the normal through nfx_brdf_spec_bwd) + the shading integral."""
        from nerfactor_amd import autograd as nfx_grad
        lxyz = self.lxyz.reshape(-1, 3)
        if self.grad_precision == 'fp32' or not self.brdf_model.tuned:
            spec = self._brdf_spec_rows(xyz, cam, normal, brdf_prop)
            return nfx_grad.ShadeSpec.apply(xyz, cam, lxyz, self.lareas,
                                            self.config.getfloat('DEFAULT', 'learned_brdf_scale'), to_srgb, normal,
                                            albedo, spec, light_vis, light)
        nets = self.brdf_model.net
        fwd_blob = self._blob128('brdf_mlp', 'brdf_out', _capi.IN_Z_RUSINK, 1, z_dim=self.z_dim, nets=nets)

        def train_blob():
            ks, bs = nets['brdf_mlp'].kernels_and_biases()
            ko, bo = nets['brdf_out'].kernels_and_biases()
            ks, bs = ks + ko, bs + bo
            return self._packed('brdf_mlp_train' + nfx_grad.GRAD_PREC, ks + bs,
                                lambda k, b: ops.pack_brdf_train_weights(k, b, self.z_dim, prec=nfx_grad.GRAD_PREC))
        lxyz = self.lxyz.reshape(-1, 3)
        spec = nfx_grad.BrdfSpec.apply(xyz, cam, lxyz, fwd_blob, train_blob, self.precision, normal, brdf_prop)
        return nfx_grad.ShadeSpec.apply(xyz, cam, lxyz, self.lareas,
                                        self.config.getfloat('DEFAULT', 'learned_brdf_scale'), to_srgb, normal,
                                        albedo, spec, light_vis, light)

    def _brdf_spec_rows(self, xyz, cam, normal, brdf_prop):
        """spec[N, L] of the frozen prior on EXPLICIT rows (nerfactor.py:413-461) — the path of grad_precision = fp32 (the
        fused bf16 kernels nfx_brdf_spec_fwd / _bwd have no fp32 instantiation) and of a prior whose shape those kernels do
        not implement.  The prior's input rows [z | embed(rusink)] of every (point, light) pair are made by ONE libnfx
        kernel (local frames, Rusinkiewicz angles and the Embedder in fp32, csrc/brdf_rows_geom.hip) and the prior's MLP
        runs on the runtime-shaped kernels (operands = `precision`; in a training call in their input-gradient mode: the
        prior is frozen); the pull-back to the normal (with the reference's custom gradients of safe_acos / safe_atan2)
        and to the BRDF code z is the twin kernel.  Back-lit rows are evaluated and multiplied by 0 instead of being
        compacted away (nerfactor.py:429-434 masks them "for speed"): no data-dependent shape, no host round trip, so the
        step can be captured in a hipGraph."""
        from nerfactor_amd import autograd as nfx_grad
        lxyz = self.lxyz.reshape(-1, 3).contiguous()
        n, nl = xyz.shape[0], lxyz.shape[0]
        prior = self.brdf_model
        nf = prior.embedder['rusink'].n_freqs
        if torch.is_grad_enabled() and (normal.requires_grad or brdf_prop.requires_grad):
            rows, front = nfx_grad.BrdfRowsGeom.apply(xyz, cam, lxyz, nf, normal, brdf_prop)
            ks, bs = prior.net['brdf_mlp'].kernels_and_biases()
            ko, bo = prior.net['brdf_out'].kernels_and_biases()
            y = nfx_grad.GenericMlp.apply(rows, lambda: prior._generic_net(train=True, prec=self.generic_prec),
                                          *(ks + ko + bs + bo))
            return (y[:, 0] * front).reshape(n, nl)
        spec = torch.empty((n, nl), dtype=torch.float32, device=xyz.device)
        per = max(1, self.mlp_chunk // nl)          # (rows of mlp_chunk / L points at a time, like chunk_apply)
        net = prior._generic_net(prec=self.generic_prec)
        for i in range(0, n, per):
            sl = slice(i, i + per)
            rows, front = ops.brdf_rows_geom_fwd(xyz[sl], cam[sl], normal[sl].detach(), brdf_prop[sl].detach(), lxyz, nf)
            y = ops.mlp_generic_fwd(rows, net)
            spec[sl] = (y[:, 0] * front).reshape(-1, nl)
        return spec

    # ------------------------------------------------------------------ loss
    def compute_loss(self, pred, gt, **kwargs):
        cfg = self.config
        normal_loss_weight = cfg.getfloat('DEFAULT', 'normal_loss_weight')
        lvis_loss_weight = cfg.getfloat('DEFAULT', 'lvis_loss_weight')
        smooth = _mae if cfg.getboolean('DEFAULT', 'smooth_use_l1') else _mse
        light_tv_weight = cfg.getfloat('DEFAULT', 'light_tv_weight')
        light_achro_weight = cfg.getfloat('DEFAULT', 'light_achro_weight')
        kwargs.pop('keep_batch', None)  # the driver always asks for per-ray losses; that is what this returns
        mode = kwargs.pop('mode')
        normal_jitter = kwargs.pop('normal_jitter')
        lvis_jitter = kwargs.pop('lvis_jitter')
        albedo_jitter = kwargs.pop('albedo_jitter')
        brdf_prop_jitter = kwargs.pop('brdf_prop_jitter')
        alpha = gt['alpha']
        bg = 1. if self.white_bg else 0.
        smooth_kind = 'mae' if smooth is _mae else 'mse'
        shape_terms = mode != 'vali' and self.shape_mode in ('scratch', 'finetune')
        if pred['rgb'].is_cuda:
            # one libnfx launch forward, one backward (loss.hip) instead of ~70 + ~80 elementwise torch launches
            tensors, spec = [], []

            def slot(t):
                for i, u in enumerate(tensors):
                    if u is t:
                        return i
                tensors.append(t)
                return len(tensors) - 1

            def term(a, b, w, kind, blend_a, blend_b):
                spec.append((slot(a), slot(b), float(w), kind, blend_a, blend_b))
            term(pred['rgb'], gt['rgb'], 1., 'mse', True, True)
            if mode != 'vali':
                if shape_terms:
                    term(pred['normal'], gt['normal'], normal_loss_weight, 'mse', True, True)
                    term(pred['lvis'], gt['lvis'], lvis_loss_weight, 'mse', True, True)
                    if normal_jitter is not None:    # the blended prediction against the raw jittered one, as upstream
                        term(pred['normal'], normal_jitter, self.normal_smooth_weight, smooth_kind, True, False)
                    if lvis_jitter is not None:
                        term(pred['lvis'], lvis_jitter, self.lvis_smooth_weight, smooth_kind, True, False)
                if albedo_jitter is not None:
                    term(pred['albedo'], albedo_jitter, self.albedo_smooth_weight, smooth_kind, False, False)
                if brdf_prop_jitter is not None:
                    term(pred['brdf'], brdf_prop_jitter, self.brdf_smooth_weight, smooth_kind, False, False)
            loss = nfx_grad.PairLoss.apply(alpha, bg, tuple(spec), *tensors)
            if mode == 'vali':
                return loss
        else:   # host tensors (the loss definition in plain torch; nothing of the render path runs there)
            def on_bg(x):
                return imgutil.alpha_blend(x, alpha, torch.full_like(x, bg))

            rgb_pred, rgb_gt = on_bg(pred['rgb']), on_bg(gt['rgb'])
            normal_pred, normal_gt = on_bg(pred['normal']), on_bg(gt['normal'])
            lvis_pred, lvis_gt = on_bg(pred['lvis']), on_bg(gt['lvis'])
            loss = _mse(rgb_gt, rgb_pred)
            if mode == 'vali':
                return loss
            if shape_terms:
                loss = loss + normal_loss_weight * _mse(normal_gt, normal_pred)
                loss = loss + lvis_loss_weight * _mse(lvis_gt, lvis_pred)
                if normal_jitter is not None:
                    loss = loss + self.normal_smooth_weight * smooth(normal_pred, normal_jitter)
                if lvis_jitter is not None:
                    loss = loss + self.lvis_smooth_weight * smooth(lvis_pred, lvis_jitter)
            if albedo_jitter is not None:
                loss = loss + self.albedo_smooth_weight * smooth(pred['albedo'], albedo_jitter)
            if brdf_prop_jitter is not None:
                loss = loss + self.brdf_smooth_weight * smooth(pred['brdf'], brdf_prop_jitter)
        if mode == 'train' and self.light.is_cuda and (light_tv_weight > 0 or light_achro_weight > 0):
            loss = loss + nfx_grad.LightSmoothness.apply(self.light, max(light_tv_weight, 0.), max(light_achro_weight, 0.))
        elif mode == 'train':
            light = self.light
            if light_tv_weight > 0:
                dx = light - torch.roll(light, 1, 1)
                dy = light - torch.roll(light, 1, 0)
                loss = loss + light_tv_weight * (dx ** 2 + dy ** 2).sum()
            if light_achro_weight > 0:
                dc = light - torch.roll(light, 1, 2)
                loss = loss + light_achro_weight * (dc ** 2).sum()
        return self.check_numerics(loss, "Loss")
