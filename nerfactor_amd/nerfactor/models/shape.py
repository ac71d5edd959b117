"""models.shape.Model — surface normals + light visibility MLPs (reference:
nerfactor/models/shape.py:33-277), evaluated by libnfx: nfx_mlp128_xyz_fwd for the normals,
nfx_lvis_fwd for the visibility of all (point, light) pairs."""
from os.path import join

import numpy as np
import torch

from nerfactor_amd import _capi, autograd as nfx_grad, ops
from nerfactor_amd.brdf.renderer import gen_light_xyz

from ..networks import mlp
from ..networks.embedder import Embedder
from ..util import img as imgutil, math as mathutil
from .base import Model as BaseModel


class Model(BaseModel):
    def __init__(self, config, debug=False):
        super().__init__(config, debug=debug)
        cfg = self.config
        self.white_bg = cfg.getboolean('DEFAULT', 'white_bg')
        self.mlp_chunk = cfg.getint('DEFAULT', 'mlp_chunk')
        self.normal_smooth_weight = cfg.getfloat('DEFAULT', 'normal_smooth_weight', fallback=0.)
        self.lvis_smooth_weight = cfg.getfloat('DEFAULT', 'lvis_smooth_weight', fallback=0.)
        self.normal_precision = cfg.get('DEFAULT', 'normal_precision', fallback='fp32')
        if self.normal_precision not in ('bf16', 'fp32'):
            raise ValueError("normal_precision = %s (bf16 | fp32)" % self.normal_precision)
        self.tuned = True
        self.embedder = self._init_embedder()
        self.net = self._init_net()
        # big world-space coordinates (e.g. MVS reconstructions) are scaled before the MLPs
        self.xyz_scale = cfg.getfloat('DEFAULT', 'xyz_scale', fallback=1.)
        lxyz, lareas = self._gen_lights()
        self.register_buffer('lxyz', lxyz)
        self.register_buffer('lareas', lareas)
        self.register_trainable()

    # ------------------------------------------------------------------ construction
    def _gen_lights(self):
        mvs_root = self.config.get('DEFAULT', 'mvs_root', fallback=None)
        if mvs_root is None:
            light_h = self.config.getint('DEFAULT', 'light_h')
            lxyz, lareas = gen_light_xyz(light_h, 2 * light_h)
        else:  # MVS initialisation ships its own light locations
            with open(join(mvs_root, 'lights.npz'), 'rb') as h:
                data = dict(np.load(h))
            lxyz, lareas = data['lxyzs'], data['lareas']
        return (torch.as_tensor(np.asarray(lxyz), dtype=torch.float32),
                torch.as_tensor(np.asarray(lareas), dtype=torch.float32))

    def _mlp128(self, in_dims, out_dims, out_act):
        cfg = self.config
        width = cfg.getint('DEFAULT', 'mlp_width')
        depth = cfg.getint('DEFAULT', 'mlp_depth')
        skip_at = cfg.getint('DEFAULT', 'mlp_skip_at')
        # self.tuned: the shipped surface MLP (mlp_width = 128, mlp_depth = 4, mlp_skip_at = 2 on 10 / 4 encoding bands)
        # runs on the tuned kernels, forward and backward.  Other shapes the reference can build (shape.py:79-94) render
        # and train through the runtime-shaped kernels (csrc/mlp_generic.hip), one autograd node per network.
        if (width, depth, skip_at) != (128, 4, 2):
            self.tuned = False
            # (a skip behind the body's LAST layer is an inner skip of the body + head network the kernels evaluate: the
            # head then reads concat(y, x), which is what body.build returns)
            if not (1 <= width <= 512 and 2 <= depth <= 12 and 0 <= skip_at < depth):
                raise NotImplementedError(
                    "libnfx's runtime-shaped kernels take mlp_width <= 512, 2 <= mlp_depth <= 12 and 0 <= mlp_skip_at < "
                    "mlp_depth (got mlp_width = %d, mlp_depth = %d, mlp_skip_at = %d)" % (width, depth, skip_at))
        body = mlp.Network([width] * depth, act=['relu'] * depth, skip_at=[skip_at])
        head = mlp.Network([out_dims], act=[out_act])
        head.build(body.build(in_dims))
        return body, head

    def _init_net(self):
        dx = self.embedder['xyz'].out_dims
        dl = self.embedder['ldir'].out_dims
        net = {}
        net['normal_mlp'], net['normal_out'] = self._mlp128(dx, 3, None)  # normalised elsewhere
        net['lvis_mlp'], net['lvis_out'] = self._mlp128(dx + dl, 1, 'sigmoid')
        return net

    def _init_embedder(self):
        cfg = self.config
        lx = cfg.getint('DEFAULT', 'n_freqs_xyz')
        ll = cfg.getint('DEFAULT', 'n_freqs_ldir')
        lv = cfg.getint('DEFAULT', 'n_freqs_vdir')
        if not cfg.getboolean('DEFAULT', 'pos_enc'):   # tf.identity in the reference (shape.py:97-106): no bands, the input itself
            lx = ll = lv = 0
        if (lx, ll) != (10, 4):
            self.tuned = False     # other band counts (0 included): the runtime-shaped path (nfx_embed takes any)
        return {name: Embedder(incl_input=True, in_dims=3, log2_max_freq=max(L - 1, 0), n_freqs=L)
                for name, L in (('xyz', lx), ('ldir', ll), ('vdir', lv))}

    # ------------------------------------------------------------------ packed weights
    def _blob128(self, body_name, head_name, in_kind, out_dim, z_dim=0, nets=None, prec=None):
        nets = self.net if nets is None else nets
        prec = self.precision if prec is None else prec
        ks, bs = nets[body_name].kernels_and_biases()
        ko, bo = nets[head_name].kernels_and_biases()
        ks, bs = ks + ko, bs + bo
        return self._packed(
            body_name + prec, ks + bs,
            lambda k, b: ops.pack_mlp128_weights(k, b, in_kind, out_dim, z_dim=z_dim, prec=prec))

    def _train_blob128(self, body_name, head_name, in_kind, out_dim, nets=None):
        """Forward + dgrad fragments for the fused backward kernel (packed lazily, cached like _blob128)."""
        nets = self.net if nets is None else nets
        ks, bs = nets[body_name].kernels_and_biases()
        ko, bo = nets[head_name].kernels_and_biases()
        ks, bs = ks + ko, bs + bo
        return self._packed(
            body_name + '_train' + nfx_grad.GRAD_PREC, ks + bs,
            lambda k, b: ops.pack_mlp128_train_weights(k, b, in_kind, out_dim, prec=nfx_grad.GRAD_PREC))

    def _params128(self, body_name, head_name, nets=None):
        nets = self.net if nets is None else nets
        ks, bs = nets[body_name].kernels_and_biases()
        ko, bo = nets[head_name].kernels_and_biases()
        return tuple(ks + ko) + tuple(bs + bo)

    @staticmethod
    def _wants_grad(params):
        return torch.is_grad_enabled() and any(p.requires_grad for p in params)

    def _mlp128_xyz(self, pts, body, head, out_dim, out_act=None, post_scale=1., post_bias=0., infer_prec=None):
        """One xyz-conditioned head; differentiable w.r.t. its weights when autograd is recording.  `infer_prec`:
        operand type of the forward-only evaluation (vali / test / render) when it differs from `precision`."""
        params = self._params128(body, head)
        if not self._net_tuned(body) or self._fp32_grads(params):
            enc = ops.embed(self.embedder['xyz'].n_freqs, x=(pts.detach() * self.xyz_scale).contiguous())
            y = self._generic_apply(enc, body, head, out_act, params)
            return y if (post_scale == 1. and post_bias == 0.) else y * post_scale + post_bias
        if self._wants_grad(params):
            blob = self._blob128(body, head, _capi.IN_XYZ, out_dim)
            return nfx_grad.Mlp128Xyz.apply(
                pts, blob, lambda: self._train_blob128(body, head, _capi.IN_XYZ, out_dim), self.precision, out_dim,
                out_act,
                self.xyz_scale, post_scale, post_bias, *params)
        prec = self.precision if infer_prec is None else infer_prec
        blob = self._blob128(body, head, _capi.IN_XYZ, out_dim, prec=prec)
        return ops.mlp128_xyz_fwd(pts, blob, out_dim, out_act=out_act, xyz_scale=self.xyz_scale,
                                  post_scale=post_scale, post_bias=post_bias, prec=prec)

    # ------------------------------------------------------------------ non-shipped shapes: runtime-shaped kernels
    def _net_tuned(self, body_name, nets=None):
        """Does THIS network have the shape the tuned kernels implement?  (A NeRFactor model may mix shapes: its shape
        networks come with the pre-trained shape model's configuration, its albedo / BRDF-code heads with its own.)"""
        body = (self.net if nets is None else nets)[body_name]
        return (self.embedder['xyz'].n_freqs == 10 and self.embedder['ldir'].n_freqs == 4 and len(body.layers) == 4 and
                all(l.units == 128 and l.activation == 'relu' for l in body.layers) and list(body.skip_at or []) == [2])

    def _fp32_grads(self, params):
        """A training call at grad_precision = fp32: every network, the shipped shapes included, runs forward and
        backward on the fp32 runtime-shaped kernels."""
        return self.grad_precision == 'fp32' and self._wants_grad(params)

    def _generic_net(self, body_name, head_name, out_act, nets=None, train=False):
        """Body + head as ONE runtime-shaped network (cached and re-packed like the tuned blobs); train = True: with the
        backward's transposed fragments."""
        nets = self.net if nets is None else nets
        body, head = nets[body_name], nets[head_name]
        ks, bs = body.kernels_and_biases()
        ko, bo = head.kernels_and_biases()
        acts = [l.activation for l in body.layers] + [out_act]
        tag = body_name + ('generic_train' if train else 'generic') + self.generic_prec
        descs = self.__dict__.setdefault('_generic_desc', {})

        blob = self._packed(tag, ks + ko + bs + bo, ops.generic_pack_fn(acts, body.skip_at, train, self.generic_prec, descs, tag))
        g = descs[tag]
        g.blob = blob
        return g

    def _generic_apply(self, rows, body_name, head_name, out_act, params, out=None):
        """net(rows): an autograd node (runtime-shaped forward + backward kernels) while the weights are being trained,
        the bare forward kernel otherwise."""
        if self._wants_grad(params):
            return nfx_grad.GenericMlp.apply(
                rows, lambda: self._generic_net(body_name, head_name, out_act, train=True), *params)
        return ops.mlp_generic_fwd(rows, self._generic_net(body_name, head_name, out_act), out=out)

    def _pred_lvis_generic(self, pts, dir_pts):
        """_pred_lvis_at (shape.py:213-237) for a non-shipped shape: per chunk of points the rows [point x light] are
        assembled as [embed(xyz_scale x) | embed(normalize(light - x_dir))] and pushed through the runtime-shaped MLP
        (mlp_chunk rows at a time, like the reference's chunk_apply)."""
        lxyz = self.lxyz.reshape(-1, 3).contiguous()
        n, L = pts.shape[0], lxyz.shape[0]
        lx, ll = self.embedder['xyz'].n_freqs, self.embedder['ldir'].n_freqs
        dx, dl = 3 + 6 * lx, 3 + 6 * ll
        params = self._params128('lvis_mlp', 'lvis_out')
        training = self._wants_grad(params)
        out = None if training else torch.empty((n, L), dtype=torch.float32, device=pts.device)
        per = max(1, self.mlp_chunk // L)
        pts = pts.detach()
        dir_pts = pts if dir_pts is None else dir_pts.detach()
        parts = []
        for i in range(0, n, per):
            x = (pts[i:i + per] * self.xyz_scale).contiguous()
            rows = torch.empty((x.shape[0] * L, dx + dl), dtype=torch.float32, device=pts.device)
            ops.embed(lx, x=x, per_ray=L, out=rows)
            ops.embed(ll, x=dir_pts[i:i + per].contiguous(), lights=lxyz, out=rows, col0=dx)
            parts.append(self._generic_apply(rows, 'lvis_mlp', 'lvis_out', 'sigmoid', params,
                                             out=None if training else out[i:i + per].view(-1, 1)))
        return torch.cat(parts, 0).view(n, L) if training else out

    # ------------------------------------------------------------------ geometry helpers
    def _calc_ldir(self, pts):
        """[N,L,3] unit directions surface -> light.  Off-path helper: the kernels recompute these
        in registers from `self.lxyz`."""
        d = self.lxyz.reshape(1, -1, 3) - pts[:, None, :]
        return mathutil.safe_l2_normalize(d, axis=2)

    @staticmethod
    def _calc_vdir(cam_loc, pts):
        return mathutil.safe_l2_normalize(cam_loc - pts, axis=1)

    @staticmethod
    def chunk_apply(func, x, dim, chunk_size):
        y = torch.zeros((x.shape[0], dim), dtype=torch.float32, device=x.device)
        for i in range(0, x.shape[0], chunk_size):
            y[i:i + chunk_size] = func(x[i:i + chunk_size])
        return y

    # ------------------------------------------------------------------ forward
    def call(self, batch, mode='train'):
        self._validate_mode(mode)
        if mode != 'train' and torch.is_grad_enabled():
            with torch.no_grad():
                return self.call(batch, mode)
        xyz_jitter_std = self.config.getfloat('DEFAULT', 'xyz_jitter_std')
        id_, hw, _, _, _, alpha, xyz, normal, lvis = batch
        xyz_noise = torch.randn_like(xyz) * xyz_jitter_std if xyz_jitter_std > 0 else None
        normal_pred = nfx_grad.l2_normalize(self._pred_normal_at(xyz))
        normal_jitter = None
        if xyz_noise is not None and self.normal_smooth_weight > 0:
            normal_jitter = nfx_grad.l2_normalize(self._pred_normal_at(xyz + xyz_noise))
        lvis_pred = self._pred_lvis_at(xyz)
        lvis_jitter = None
        if xyz_noise is not None and self.lvis_smooth_weight > 0:
            lvis_jitter = self._pred_lvis_at(xyz + xyz_noise, dir_pts=xyz)
        pred = {'normal': normal_pred, 'lvis': lvis_pred}
        gt = {'normal': normal, 'lvis': lvis, 'alpha': alpha}
        loss_kwargs = {'normal_jitter': normal_jitter, 'lvis_jitter': lvis_jitter}
        to_vis = {'id': id_, 'hw': hw}
        for k, v in pred.items():
            to_vis['pred_' + k] = v
        for k, v in gt.items():
            to_vis['gt_' + k] = v
        return pred, gt, loss_kwargs, to_vis

    def _pred_normal_at(self, pts, eps=1e-6):
        """Raw (un-normalised) normals, +eps so an all-zero prediction cannot break the tangents.

        Rendering evaluates this head with fp32-class operands (bf16 hi / lo pairs, mlp128_x3.hip) even when
        `precision = bf16` (ini key `normal_precision`, default fp32): the microfacet BRDF divides by
        4 |l.n| |v.n| (microfacet.py:57), which amplifies a bf16-sized error of the normal without bound towards
        grazing directions — 0.37 max-abs on the rendered 800 x 800 frame in round 2 — while the head is 0.2 % of a
        render's matrix work.  A training step (autograd recording) keeps the bf16 forward its backward kernel
        re-computes, so the gradient stays the exact derivative of the function the loss saw."""
        return self._mlp128_xyz(pts, 'normal_mlp', 'normal_out', 3, out_act=None, post_bias=eps,
                                infer_prec=self.normal_precision)

    def _pred_lvis_at(self, pts, surf2l=None, dir_pts=None):
        """[N,L] visibility of every light from every point.  Directions are recomputed in the
        kernel from `self.lxyz` and `dir_pts` (default `pts`); an explicit `surf2l` tensor is
        accepted for signature compatibility only when it equals _calc_ldir(dir_pts)."""
        params = self._params128('lvis_mlp', 'lvis_out')
        if self._net_tuned('lvis_mlp') and self._fp32_grads(params) and self.generic_prec == 'fp32':
            # the shipped network at grad_precision = fp32 / fp32_matrix = pairs: tuned fp32-class forward kernel, runtime-shaped
            # fp32-class backward on rows rebuilt in the backward (autograd.LvisFp32Class)
            lvis = nfx_grad.LvisFp32Class.apply(
                pts.detach(), (pts if dir_pts is None else dir_pts).detach(), self.lxyz.reshape(-1, 3).contiguous(),
                self._blob128('lvis_mlp', 'lvis_out', _capi.IN_XYZ_LDIR, 1, prec='fp32'),
                lambda: self._generic_net('lvis_mlp', 'lvis_out', 'sigmoid', train=True), self.xyz_scale,
                (self.embedder['xyz'].n_freqs, self.embedder['ldir'].n_freqs), max(self.mlp_chunk, 1 << 20), *params)
            return self.check_numerics(lvis, "Light visibility")
        if not self._net_tuned('lvis_mlp') or self._fp32_grads(params):
            return self.check_numerics(self._pred_lvis_generic(pts, dir_pts), "Light visibility")
        blob = self._blob128('lvis_mlp', 'lvis_out', _capi.IN_XYZ_LDIR, 1)
        lxyz = self.lxyz.reshape(-1, 3)
        if self._wants_grad(params):
            lvis = nfx_grad.Lvis.apply(
                pts, pts if dir_pts is None else dir_pts, lxyz, blob,
                lambda: self._train_blob128('lvis_mlp', 'lvis_out', _capi.IN_XYZ_LDIR, 1), self.precision,
                self.xyz_scale,
                *params)
        else:
            lvis = ops.lvis_fwd(pts, lxyz, blob, xyz_scale=self.xyz_scale, xyz_dir=dir_pts,
                                prec=self.precision)
        return self.check_numerics(lvis, "Light visibility")

    def _lvis_rows_ok(self):
        """May a render store its visibilities straight into the rows of a full-size buffer (round 6: ops.lvis_fwd(out=,
        out_row=) — the zero-filled scatter and the check_numerics pass of the [n, 512] tensor done by the kernel's own
        stores)?  The shipped network, bf16 operands, nothing being differentiated."""
        if torch.is_grad_enabled() or not self._net_tuned('lvis_mlp'):
            return False
        return ops.lvis_rows_supported(self.precision)

    def _pred_lvis_rows(self, pts, out, out_row, dir_pts=None):
        """_pred_lvis_at whose result for point i lands in out[out_row[i]] (out: [n_all, L], the other rows are the caller's)."""
        blob = self._blob128('lvis_mlp', 'lvis_out', _capi.IN_XYZ_LDIR, 1)
        flag = torch.zeros(1, dtype=torch.int32, device=pts.device)
        ops.lvis_fwd(pts, self.lxyz.reshape(-1, 3), blob, xyz_scale=self.xyz_scale, xyz_dir=dir_pts, prec=self.precision,
                     out=out, out_row=out_row, nan_flag=flag)
        self.check_flag(flag == 0, "Light visibility")
        return out

    # ------------------------------------------------------------------ loss
    def compute_loss(self, pred, gt, **kwargs):
        cfg = self.config
        normal_loss_weight = cfg.getfloat('DEFAULT', 'normal_loss_weight')
        lvis_loss_weight = cfg.getfloat('DEFAULT', 'lvis_loss_weight')
        smooth = _mae if cfg.getboolean('DEFAULT', 'smooth_use_l1') else _mse
        kwargs.pop('keep_batch', None)
        normal_jitter = kwargs.pop('normal_jitter')
        lvis_jitter = kwargs.pop('lvis_jitter')
        alpha = gt['alpha']
        bg = 1. if self.white_bg else 0.
        if pred['normal'].is_cuda:   # one libnfx launch forward, one backward (loss.hip)
            kind = 'mae' if smooth is _mae else 'mse'
            tensors = [pred['normal'], gt['normal'], pred['lvis'], gt['lvis']]
            spec = [(0, 1, normal_loss_weight, 'mse', True, True), (2, 3, lvis_loss_weight, 'mse', True, True)]
            if normal_jitter is not None:
                tensors.append(normal_jitter)
                spec.append((0, len(tensors) - 1, self.normal_smooth_weight, kind, True, False))
            if lvis_jitter is not None:
                tensors.append(lvis_jitter)
                spec.append((2, len(tensors) - 1, self.lvis_smooth_weight, kind, True, False))
            loss = nfx_grad.PairLoss.apply(alpha, bg, tuple(spec), *tensors)
        else:                        # host tensors: the loss definition in plain torch
            normal_pred = imgutil.alpha_blend(pred['normal'], alpha, torch.full_like(gt['normal'], bg))
            normal_gt = imgutil.alpha_blend(gt['normal'], alpha, torch.full_like(gt['normal'], bg))
            lvis_pred = imgutil.alpha_blend(pred['lvis'], alpha, torch.full_like(gt['lvis'], bg))
            lvis_gt = imgutil.alpha_blend(gt['lvis'], alpha, torch.full_like(gt['lvis'], bg))
            loss = normal_loss_weight * _mse(normal_gt, normal_pred) + \
                lvis_loss_weight * _mse(lvis_gt, lvis_pred)
            if normal_jitter is not None:
                loss = loss + self.normal_smooth_weight * smooth(normal_pred, normal_jitter)
            if lvis_jitter is not None:
                loss = loss + self.lvis_smooth_weight * smooth(lvis_pred, lvis_jitter)
        return self.check_numerics(loss, "Loss")

    # ------------------------------------------------------------------ vis (raw dumps only)


def _mse(a, b):
    return ((a - b) ** 2).mean(-1)   # keras.losses.MSE: mean over the last axis


def _mae(a, b):
    return (a - b).abs().mean(-1)
