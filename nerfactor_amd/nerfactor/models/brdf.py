"""models.brdf.Model — the MERL BRDF prior: a latent-code-conditioned MLP on Rusinkiewicz
coordinates (reference: nerfactor/models/brdf.py:34-136).  NeRFactor evaluates its frozen
`brdf_mlp`/`brdf_out` per (surface point, light) inside nfx_brdf_spec_fwd; this class owns the
weights, the latent codes and the config surface, and trains on the same fused template (row f-4)."""
from os.path import basename
import glob

import torch

from ... import autograd as nfx_grad, ops
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
        # self.tuned: the shipped prior (config/brdf.ini: 128 x 4, skip at 2, 2 encoding bands) runs on the fused
        # width-128 template, forward and backward; every other shape the reference can build (brdf.py:57-86) on the
        # runtime-shaped kernels (csrc/mlp_generic.hip), rows = [z | embed(rusink)] assembled explicitly.
        self.tuned = (width, depth, skip_at) == (128, 4, 2) and self.embedder['rusink'].n_freqs == 2
        if not self.tuned:
            # (a skip behind the body's last layer is an inner skip of the body + head network the kernels evaluate)
            if not (1 <= width <= 512 and 2 <= depth <= 12 and 0 <= skip_at < depth):
                raise NotImplementedError(
                    "libnfx's runtime-shaped kernels take mlp_width <= 512, 2 <= mlp_depth <= 12 and 0 <= mlp_skip_at < "
                    "mlp_depth (got mlp_width = %d, mlp_depth = %d, mlp_skip_at = %d)" % (width, depth, skip_at))
        body = mlp.Network([width] * depth, act=['relu'] * depth, skip_at=[skip_at])
        head = mlp.Network([1], act=['softplus'])  # reflectance > 0
        head.build(body.build(self.z_dim + self.embedder['rusink'].out_dims))
        return {'brdf_mlp': body, 'brdf_out': head}

    def _init_embedder(self):
        cfg = self.config
        n_freqs = cfg.getint('DEFAULT', 'n_freqs')
        if not cfg.getboolean('DEFAULT', 'pos_enc'):   # tf.identity in the reference (brdf.py:72-74): no bands
            n_freqs = 0
        return {'rusink': Embedder(incl_input=True, in_dims=3, log2_max_freq=max(n_freqs - 1, 0), n_freqs=n_freqs)}

    def _train_blob(self):
        """Forward + dgrad + input-gradient fragments of the prior (cached, re-packed on the device after a step)."""
        ks, bs = self.net['brdf_mlp'].kernels_and_biases()
        ko, bo = self.net['brdf_out'].kernels_and_biases()
        return self._packed('brdf_rows' + nfx_grad.GRAD_PREC, ks + ko + bs + bo,
                            lambda k, b: ops.pack_brdf_train_weights(k, b, self.z_dim, prec=nfx_grad.GRAD_PREC))

    def _eval_brdf_at(self, z, rusink):
        """Explicit-row evaluation (z [M, z_dim], rusink [M, 3]) -> (brdf, brdf_reci) [M, 1] (brdf.py:57-66, 101-106):
        one fused libnfx launch for both halves (nfx_brdf_rows_fwd); under autograd the backward is one
        nfx_brdf_rows_bwd (dgrad chain, d z, weight-gradient GEMMs).  `mlp_chunk` is not needed: nothing of size
        M x width is materialised in the forward."""
        ks, bs = self.net['brdf_mlp'].kernels_and_biases()
        ko, bo = self.net['brdf_out'].kernels_and_biases()
        params = tuple(ks + ko) + tuple(bs + bo)
        z, rusink = z.float().contiguous(), rusink.float().contiguous()
        recording = torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in params))
        if not self.tuned or (recording and self.grad_precision == 'fp32'):
            return self._eval_brdf_generic(z, rusink, params)
        if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in params)):
            brdf, reci = nfx_grad.BrdfRows.apply(z, rusink, self._train_blob, self.precision, *params)
        else:
            out = ops.brdf_rows_fwd(z, rusink, self._train_blob(), reci=True, prec=nfx_grad.GRAD_PREC)   # rows kernels: bf16 only
            brdf, reci = out[:z.shape[0]], out[z.shape[0]:]
        return brdf[:, None], reci[:, None]

    def _generic_net(self, train=False, prec=None):
        prec = self.generic_prec if prec is None else prec
        ks, bs = self.net['brdf_mlp'].kernels_and_biases()
        ko, bo = self.net['brdf_out'].kernels_and_biases()
        body = self.net['brdf_mlp']
        acts = [l.activation for l in body.layers] + ['softplus']
        tag = ('brdf_generic_train' if train else 'brdf_generic') + prec
        descs = self.__dict__.setdefault('_generic_desc', {})

        blob = self._packed(tag, ks + ko + bs + bo, ops.generic_pack_fn(acts, body.skip_at, train, prec, descs, tag))
        g = descs[tag]
        g.blob = blob
        return g

    def _eval_brdf_generic(self, z, rusink, params):
        """_eval_brdf_at for a non-shipped shape: the 2 M rows [z | embed(rusink)], [z | embed(rusink with phi_d + pi)]
        through the runtime-shaped MLP — an autograd node (weights, and z through the concatenation) when recording."""
        import math
        nf, m = self.embedder['rusink'].n_freqs, z.shape[0]
        reci = torch.cat((rusink[:, :1] + math.pi, rusink[:, 1:]), 1)
        enc = ops.embed(nf, x=torch.cat((rusink, reci), 0).contiguous())
        rows = torch.cat((torch.cat((z, z), 0), enc), 1)
        if torch.is_grad_enabled() and (z.requires_grad or any(p.requires_grad for p in params)):
            y = nfx_grad.GenericMlp.apply(rows, lambda: self._generic_net(train=True), *params)
        else:
            y = ops.mlp_generic_fwd(rows, self._generic_net())
        return y[:m], y[m:]

    # ------------------------------------------------------------------ training of the prior (brdf.py:87-136)
    def call(self, batch, mode='train'):
        """batch = (id_, i, envmap_h, ims, spp, rusink[N,3], refl[N,1]) of datasets/brdf_merl.py.  The 18-input
        128-wide MLP runs — forward, backward and weight gradients — on the fused width-128 template of libnfx
        (nfx_brdf_rows_fwd / nfx_brdf_rows_bwd), like the surface MLPs; the latent codes get their gradient from the
        kernel's d z through the gather below."""
        self._validate_mode(mode)
        if mode != 'train' and torch.is_grad_enabled():
            with torch.no_grad():
                return self.call(batch, mode=mode)
        id_, i, envmap_h, ims, spp, rusink, refl = batch
        if mode == 'test' and int(i[0]) == -1:   # novel identity: "<n>_<w1>_<mat1>_<w2>_<mat2>" (merl names may
            _, w1, rest = id_[0].split('_', 2)    # contain '-', never '_' followed by a float: split from both ends)
            mat1, w2, mat2 = self._split_interp_id(rest)
            z = self.latent_code.interp(float(w1), self.brdf_names.index(mat1), float(w2),
                                        self.brdf_names.index(mat2))
            z = z.reshape(1, -1).expand(rusink.shape[0], -1)
        else:
            z = self.latent_code(i)
        brdf, brdf_reci = self._eval_brdf_at(z, rusink)
        pred = {'brdf': brdf, 'brdf_reci': brdf_reci}
        gt = {'brdf': refl}
        to_vis = {'id': id_, 'i': i, 'z': z, 'gt_brdf': refl, 'envmap_h': envmap_h, 'ims': ims, 'spp': spp}
        to_vis.update(pred)
        return pred, gt, {}, to_vis

    def _split_interp_id(self, rest):
        """'<mat1>_<w2>_<mat2>' -> (mat1, w2, mat2) with material names that may themselves contain '_'."""
        for name in sorted(self.brdf_names, key=len, reverse=True):
            if rest.startswith(name + '_'):
                w2, mat2 = rest[len(name) + 1:].split('_', 1)
                return name, w2, mat2
        raise ValueError("cannot parse interpolation id %r" % rest)

    def compute_loss(self, pred, gt, **kwargs):
        transform = self.config.get('DEFAULT', 'loss_transform')
        if transform.lower() == 'none':
            f = lambda x: x
        elif transform == 'log':
            f = torch.log
        elif transform == 'divide':
            f = lambda x: x / (x + 1.)
        else:
            raise NotImplementedError(transform)
        loss = 0
        for weight, fn in self.wloss:   # the reciprocal Rusinkiewicz coordinates share the ground truth
            loss = loss + weight * fn(f(gt['brdf']), f(pred['brdf']), **kwargs)
            loss = loss + weight * fn(f(gt['brdf']), f(pred['brdf_reci']), **kwargs)
        return loss

    def vis_batch(self, data_dict, outdir, mode='train', dump_raw_to=None, **kwargs):
        """Raw dump only (the reference renders the BRDF on spheres here, brdf.py:138-220: viewer tooling)."""
        import os
        import numpy as np
        self._validate_mode(mode)
        if dump_raw_to is None and mode == 'train':
            return
        path = dump_raw_to if dump_raw_to is not None else os.path.join(outdir, 'raw.npz')
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        arrays = {k: (v.detach().float().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v))
                  for k, v in data_dict.items() if v is not None}
        with open(path, 'wb') as h:
            np.savez(h, **arrays)
        if mode != 'train':
            os.makedirs(outdir, exist_ok=True)
