"""models.nerf.Model — vanilla NeRF with hierarchical sampling (reference:
nerfactor/models/nerf.py:33-300), ray marching on libnfx:

    rayd normalise -> gen_z -> [fused pts + posenc + MLP] -> composite -> sample_fine
                   -> [fused pts + posenc + MLP (fine net)] -> composite

`call` / `compute_loss` / the statics keep the reference's signatures and return structures.
"""
import torch

from nerfactor_amd import autograd, autograd as nfx_grad, ops

from .. import losses
from ..networks import mlp
from ..networks.embedder import Embedder
from .base import Model as BaseModel


class Model(BaseModel):
    DEFAULT_FP32_MATRIX = 'native'      # (models/base.py: where `pairs` leaves the stated 1e-3 of the reference's gradients)

    def __init__(self, config, debug=False):
        super().__init__(config, debug=debug)
        cfg = self.config
        self.use_views = cfg.getboolean('DEFAULT', 'use_views')
        self.near = cfg.getfloat('DEFAULT', 'near')
        self.far = cfg.getfloat('DEFAULT', 'far')
        self.n_samples_fine = cfg.getint('DEFAULT', 'n_samples_fine')
        self.white_bg = cfg.getboolean('DEFAULT', 'white_bg')
        self.embedder = self._init_embedder()
        self.net = {}
        stages = ('coarse', 'fine') if self.n_samples_fine > 0 else ('coarse',)
        for stage in stages:
            for k, v in self._init_net().items():
                self.net[stage + '_' + k] = v
        self._check_fusable()
        # The last sample of a ray has dist = 1e10 (nerf.py:186-191), so alpha_last = [sigma_last > 0] EXACTLY: a ray
        # whose sigma_last sits within the bf16 kernel's rounding of 0 flips between "hits the far plane" and
        # "background", and no tolerance on rgb can hold for it.  When rendering (autograd off) with precision = bf16
        # the density of every ray's LAST sample is therefore re-evaluated by the fp32-class density kernel
        # (nerf_geom_x3.hip: 1 of n_samples points at ~3x the cost; ini key `last_sample_precision`, default fp32).
        # A training step keeps the bf16 forward its backward kernel re-computes.
        self.last_sample_precision = self.config.get('DEFAULT', 'last_sample_precision', fallback='fp32')
        if self.last_sample_precision not in ('bf16', 'fp32'):
            raise ValueError("last_sample_precision = %s (bf16 | fp32)" % self.last_sample_precision)
        # `coarse_precision` (render-time, precision = bf16 only; see _eval_rays and _coarse_refine_on):
        #   bf16    the coarse pass as the bf16 kernel computes it;
        #   select  + the fp32-class density for the coarse samples that decide where the fine samples go
        #           (ops.nerf_refine_coarse: visible and unsaturated or sign-undecided; ~9 % of the coarse samples of a fitted scene);
        #   auto    (default) `select` when the bf16 density error of THESE weights can move a sample's alpha by more than
        #           `coarse_refine_gate` (measured once per weight version), `bf16` otherwise;
        #   fp32    the whole coarse pass fp32-class (25 % of a frame's points at 3.7x).
        self.coarse_precision = self.config.get('DEFAULT', 'coarse_precision', fallback='auto')
        if self.coarse_precision not in ('auto', 'bf16', 'select', 'fp32'):
            raise ValueError("coarse_precision = %s (auto | bf16 | select | fp32)" % self.coarse_precision)
        self.coarse_refine_gate = self.config.getfloat('DEFAULT', 'coarse_refine_gate', fallback=4e-3)
        self._coarse_gate = None
        self.register_trainable()

    # ------------------------------------------------------------------ construction
    def _init_net(self):
        cfg = self.config
        width = cfg.getint('DEFAULT', 'mlp_width')
        depth = cfg.getint('DEFAULT', 'enc_depth')
        act = cfg.get('DEFAULT', 'act', fallback='relu')
        dx = self.embedder['xyz'].out_dims
        enc = mlp.Network([width] * depth, act=[act] * depth, skip_at=[depth // 2])
        # (enc_depth = 2: the skip sits behind the LAST layer, so the encoder's output is concat(y, embed(x)) and every head
        # reads width + dx features — nerf.py:53-71 with mlp.py:47-48; Keras infers the heads' input width)
        d_enc = enc.build(dx)
        net = {'enc': enc}
        if not self.use_views:
            net['rgbs_out'] = mlp.Network([4], act=[None])
            net['rgbs_out'].build(d_enc)
            return net
        dv = self.embedder['view'].out_dims
        net['sigma_out'] = mlp.Network([1], act=[None])       # relu applied when compositing
        net['bottleneck'] = mlp.Network([width], act=[None])
        net['rgb_out'] = mlp.Network([width // 2, 3], act=[act, None])  # sigmoid when compositing
        net['sigma_out'].build(d_enc)
        net['bottleneck'].build(d_enc)
        net['rgb_out'].build(width + dv)
        return net

    def _init_embedder(self):
        cfg = self.config
        lx = cfg.getint('DEFAULT', 'n_freqs_xyz')
        lv = cfg.getint('DEFAULT', 'n_freqs_view')
        if not cfg.getboolean('DEFAULT', 'pos_enc'):   # tf.identity in the reference (nerf.py:81-85): n_freqs = 0
            lx = lv = 0
        elif not self.use_views:                        # nerf.py:103-104: the view embedder is the identity
            lv = 0
        return {
            'xyz': Embedder(incl_input=True, in_dims=3, log2_max_freq=max(lx - 1, 0), n_freqs=lx),
            'view': Embedder(incl_input=True, in_dims=3, log2_max_freq=max(lv - 1, 0), n_freqs=lv)}

    def _check_fusable(self):
        """self.tuned: the shipped architecture (config/nerf.ini: mlp_width = 256, enc_depth = 8, relu, use_views,
        n_freqs_xyz = 10, n_freqs_view = 4) runs on the tuned kernels, forward and backward.  Every other shape the
        reference can build (nerf.py:53-90: other widths and depths, use_views = False, pos_enc = False, other band
        counts) renders, trains and yields its geometry (eval_sigma / eval_sigma_normal) through the runtime-shaped
        kernels (csrc/mlp_generic.hip: nfx_embed, nfx_mlp_generic_fwd / _bwd, bf16 operands like the tuned path), one
        autograd node per network."""
        cfg = self.config
        width, depth = cfg.getint('DEFAULT', 'mlp_width'), cfg.getint('DEFAULT', 'enc_depth')
        act = cfg.get('DEFAULT', 'act', fallback='relu')
        self.tuned = (self.use_views and width == 256 and depth == 8 and act == 'relu' and
                      self.embedder['xyz'].n_freqs == 10 and self.embedder['view'].n_freqs == 4)
        if self.tuned:
            return
        if act not in (None, 'relu', 'sigmoid', 'softplus') or depth < 2 or depth > 12 or not 1 <= width <= 512:
            raise NotImplementedError(
                "libnfx's runtime-shaped kernels take 2 <= enc_depth <= 12, mlp_width <= 512 and relu / sigmoid / softplus "
                "/ linear activations (got enc_depth = %d, mlp_width = %d, act = %s)" % (depth, width, act))

    # ------------------------------------------------------------------ weights -> device blob
    def _nerf_blob(self, pref):
        ks, bs = self._nerf_params(pref)
        return self._packed(pref + self.precision, ks + bs,
                            lambda k, b: ops.pack_nerf_weights(k, b, self.precision))

    # ------------------------------------------------------------------ forward
    def call(self, batch, mode='train'):
        self._validate_mode(mode)
        if mode != 'train' and torch.is_grad_enabled():
            with torch.no_grad():
                return self.call(batch, mode=mode)
        id_, hw, rayo, rayd, rgb = batch
        pred_coarse, pred_fine = self._render_rays(rayo, rayd, mode=mode)
        pred = {'coarse': pred_coarse['rgb'], 'fine': pred_fine.get('rgb', None)}
        to_vis = {'id': id_, 'hw': hw, 'gt_rgb': rgb}
        for k, v in pred_coarse.items():
            to_vis['coarse_' + k] = v
        for k, v in pred_fine.items():
            to_vis['fine_' + k] = v
        return pred, rgb, {}, to_vis

    @staticmethod
    def gen_z(near, far, n_samples, n_rays, lin_in_disp=False, perturb=False, device='cuda'):
        u = torch.rand((n_rays, n_samples), device=device) if perturb else None
        return ops.gen_z(near, far, n_samples, n_rays, lin_in_disp=lin_in_disp, u=u, device=device)

    @staticmethod
    def gen_z_fine(z_coarse, weights, n_samples_fine, perturb=False):
        u = None
        if perturb:
            u = torch.rand((z_coarse.shape[0], n_samples_fine), device=z_coarse.device)
        return ops.sample_fine(z_coarse, weights.detach(), n_samples_fine, u=u)

    @staticmethod
    def accumulate_sigma(sigma, z, rayd, noise_std=0., inf=1e10, accu_chunk=65536):
        """weights[N,S] from densities; `accu_chunk` is accepted for signature compatibility —
        the kernel needs no chunking.  `inf` (nerf.py:186-191: the distance the LAST sample is given) is 1e10 inside the
        compositing kernel; another value is evaluated through the same kernel on S + 1 samples: an empty sample
        (density 0: alpha = 0, weight 0) placed `inf` behind the last one gives that one dist = inf, and is dropped."""
        n, s = sigma.shape
        noise = torch.randn_like(sigma) * noise_std if noise_std > 0 else None
        extra = inf != 1e10
        if extra:
            z = torch.cat((z, z[:, -1:] + float(inf)), 1)
            sigma = torch.cat((sigma, torch.zeros_like(sigma[:, :1])), 1)
            if noise is not None:
                noise = torch.cat((noise, torch.zeros_like(noise[:, :1])), 1)
        rgbs = torch.zeros((n, s + extra, 4), dtype=torch.float32, device=sigma.device)
        rgbs[:, :, 3] = sigma
        w = ops.composite_fwd(rgbs, z.contiguous(), rayd, white_bg=False, noise=noise)[4]
        return w[:, :s].contiguous() if extra else w

    def _nerf_params(self, pref):
        nets = [self.net[pref + k] for k in ('enc', 'sigma_out', 'bottleneck', 'rgb_out')]
        ks, bs = [], []
        for n in nets:
            k, b = n.kernels_and_biases()
            ks += k
            bs += b
        return ks, bs

    def _nerf_train_blob(self, pref):
        ks, bs = self._nerf_params(pref)
        return self._packed(pref + 'train' + nfx_grad.GRAD_PREC, ks + bs,
                            lambda k, b: ops.pack_nerf_train_weights(k, b, nfx_grad.GRAD_PREC))

    # ------------------------------------------------------------------ non-shipped shapes: runtime-shaped kernels
    def _generic_net(self, key, train=False):
        """self.net[key] packed for nfx_mlp_generic_fwd / _bwd (cached like the tuned blobs, re-packed — on the device —
        when a parameter changes); train = True: with the backward's transposed fragments.  Operand type = `precision`."""
        net = self.net[key]
        ks, bs = net.kernels_and_biases()
        acts = [l.activation for l in net.layers]
        tag = key + ('generic_train' if train else 'generic') + self.generic_prec
        descs = self.__dict__.setdefault('_generic_desc', {})

        inner_skips = None if net.skip_at is None else [i for i in net.skip_at if i < len(net.layers) - 1]    # (a trailing one: _enc_out)
        blob = self._packed(tag, ks + bs, ops.generic_pack_fn(acts, inner_skips, train, self.generic_prec, descs, tag))
        g = descs[tag]
        g.blob = blob
        return g

    def _enc_out(self, pref, feat, emb):
        """The encoder's output given its last layer's: a skip behind the LAST layer (enc_depth = 2: skip_at = [1]) appends the
        embedded input, y first (mlp.py:47-48)."""
        enc = self.net[pref + 'enc']
        if enc.skip_at is not None and len(enc.layers) - 1 in enc.skip_at:
            return torch.cat((feat, emb), 1)
        return feat

    def _generic_nets(self, pref, train=False):
        names = ('enc', 'sigma_out', 'bottleneck', 'rgb_out') if self.use_views else ('enc', 'rgbs_out')
        return {n: self._generic_net(pref + n, train) for n in names}

    def _generic_apply(self, key, x):
        """net(x) recorded for autograd: forward and backward on the runtime-shaped kernels."""
        ks, bs = self.net[key].kernels_and_biases()
        return autograd.GenericMlp.apply(x, lambda: self._generic_net(key, train=True), *(ks + bs))

    def _eval_rays_generic_train(self, rayo, rayd, z, pref):
        """_eval_rays_generic with every network an autograd node (the embeddings are data): what trainvali.py's
        GradientTape differentiates for a non-shipped shape."""
        n, s = z.shape
        lx, lv = self.embedder['xyz'].n_freqs, self.embedder['view'].n_freqs
        emb = ops.embed(lx, rayo=rayo, rayd=rayd, z=z)
        feat = self._enc_out(pref, self._generic_apply(pref + 'enc', emb), emb)
        if not self.use_views:
            return self._generic_apply(pref + 'rgbs_out', feat).view(n, s, 4)
        sigma = self._generic_apply(pref + 'sigma_out', feat)
        bott = self._generic_apply(pref + 'bottleneck', feat)
        cat = torch.cat((bott, ops.embed(lv, rayd=rayd, per_ray=s)), 1)
        rgb = self._generic_apply(pref + 'rgb_out', cat)
        return torch.cat((rgb, sigma), 1).view(n, s, 4)

    def _eval_rays_generic(self, rayo, rayd, z, pref):
        """_eval_nerf_at (nerf.py:256-290) on the runtime-shaped kernels, chunked by `mlp_chunk` points like the
        reference: embed(o + d z) -> enc -> [sigma_out | bottleneck -> concat(embed(view)) -> rgb_out] or rgbs_out."""
        nets = self._generic_nets(pref)
        n, s = z.shape
        lx, lv = self.embedder['xyz'].n_freqs, self.embedder['view'].n_freqs
        rays_per_chunk = max(1, self.config.getint('DEFAULT', 'mlp_chunk') // s)
        rgbs = torch.empty((n, s, 4), dtype=torch.float32, device=z.device)
        for r0 in range(0, n, rays_per_chunk):
            r1 = min(n, r0 + rays_per_chunk)
            o, d, zz = rayo[r0:r1].contiguous(), rayd[r0:r1].contiguous(), z[r0:r1].contiguous()
            emb = ops.embed(lx, rayo=o, rayd=d, z=zz)
            feat = self._enc_out(pref, ops.mlp_generic_fwd(emb, nets['enc']), emb)
            out = rgbs[r0:r1].view(-1, 4)
            if not self.use_views:
                ops.mlp_generic_fwd(feat, nets['rgbs_out'], out=out)
                continue
            w = nets['bottleneck'].d_out
            cat = torch.empty((feat.shape[0], w + 3 + 6 * lv), dtype=torch.float32, device=z.device)
            ops.mlp_generic_fwd(feat, nets['bottleneck'], out=cat)              # columns [0, w)
            ops.embed(lv, rayd=d, per_ray=s, out=cat, col0=w)                   # columns [w, w + 3 + 6 lv)
            ops.mlp_generic_fwd(cat, nets['rgb_out'], out=out)                  # rgb -> columns 0..2
            ops.mlp_generic_fwd(feat, nets['sigma_out'], out=out, col0=3)       # sigma -> column 3
        return rgbs

    def _eval_rays(self, rayo, rayd, z, pref):
        """rgbs[N,S,4]; differentiable w.r.t. the network weights while autograd is recording."""
        training = torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters())
        if training and (not self.tuned or self.grad_precision == 'fp32'):
            return self._eval_rays_generic_train(rayo, rayd, z, pref)     # (fp32 gradients: the fp32 runtime-shaped kernels)
        if not self.tuned:
            return self._eval_rays_generic(rayo, rayd, z, pref)
        if torch.is_grad_enabled():
            ks, bs = self._nerf_params(pref)
            return autograd.NerfMlp.apply(rayo, rayd, z, self._nerf_blob(pref),
                                          lambda: self._nerf_train_blob(pref), self.precision, *(ks + bs))
        if pref == 'coarse_' and self.precision == 'bf16' and self.coarse_precision == 'fp32':
            # (render-time option, round 5: the COARSE pass with fp32-class operands — its weights place the fine samples,
            #  and on a fitted network's sharp density edge bf16 coarse weights move a silhouette ray's fine samples by up
            #  to 1.5 coarse bins, DESIGN.md section 3.4 / profiles/HISTORY.md section 4; 25 % of a frame's points at 3.7x)
            return ops.nerf_mlp_fwd(rayo, rayd, z, self._packed(
                pref + 'fp32', sum(self._nerf_params(pref), []), lambda k, b: ops.pack_nerf_weights(k, b, 'fp32')), 'fp32')
        rgbs = ops.nerf_mlp_fwd(rayo, rayd, z, self._nerf_blob(pref), self.precision)
        if self.precision == 'bf16' and self.last_sample_precision == 'fp32' and z.shape[1] > 1:
            ops.nerf_refine_last_sample(rayo, rayd, z, rgbs, self._nerf_geom_blob(pref, 'fp32'))
        if pref == 'coarse_' and self.precision == 'bf16' and self.n_samples_fine > 0 and self._coarse_refine_on(rayo, rayd, z, rgbs):
            ops.nerf_refine_coarse(rayo, rayd, z, rgbs, self._nerf_geom_blob(pref, 'fp32'),
                                   sigma_margin=ops.REFINE_MARGIN_FACTOR * self._coarse_gate[3])
        return rgbs

    def _coarse_refine_on(self, rayo, rayd, z, rgbs):
        """Does this render re-evaluate the deciding coarse samples fp32-class (round 6)?  The inverse-CDF sampler
        (util/math.py:71-94) is chaotic on the silhouette rays of a FITTED network: bf16 density errors of 0.1-0.3 (2 % of a
        sample's alpha) move fine samples across the density edge, 0.4 % of the rays of a fitted view end up 3e-2 .. 0.23 from the
        fp32 render (scripts/coarse_refine_probe.py: 509 of 131 072 rays; 0 with the refinement).  On weights whose bf16
        error is ten times smaller (glorot initialisation: 1.5e-3 of alpha) nothing is gained, and every sample of such a
        fog qualifies — so `auto` measures the error of the weights at hand, once per weight version."""
        if self.coarse_precision not in ('auto', 'select') or z.shape[0] == 0 or z.shape[1] < 3:
            return False
        versions = tuple((t.data_ptr(), t._version) for t in sum(self._nerf_params('coarse_'), []))
        if self._coarse_gate is None or self._coarse_gate[0] != versions:    # (weights, decision, alpha error, sigma error)
            err_a, err_s = ops.nerf_coarse_error(rayo, rayd, z, rgbs, self._nerf_geom_blob('coarse_', 'fp32'))
            self._coarse_gate = (versions, err_a > self.coarse_refine_gate, err_a, err_s)
        return self.coarse_precision == 'select' or self._coarse_gate[1]

    def _render_rays(self, rayo, rayd, mode='train'):
        cfg = self.config
        n_coarse = cfg.getint('DEFAULT', 'n_samples_coarse')
        lin_in_disp = cfg.getboolean('DEFAULT', 'lin_in_disp')
        perturb = cfg.getboolean('DEFAULT', 'perturb') if mode == 'train' else False
        rayd = ops.l2_normalize3(rayd, 1e-12)
        z = self.gen_z(self.near, self.far, n_coarse, rayo.shape[0], lin_in_disp=lin_in_disp,
                       perturb=perturb, device=rayo.device)
        rgbs = self._eval_rays(rayo, rayd, z, 'coarse_')
        rgb, occu, depth, disp, weights = self._accumulate(rgbs, z, rayd)
        pred_coarse = {'rgb': rgb, 'occu': occu, 'depth': depth, 'disp': disp}
        if self.n_samples_fine <= 0:
            return pred_coarse, {}
        z = self.gen_z_fine(z, weights, self.n_samples_fine, perturb=perturb)
        rgbs = self._eval_rays(rayo, rayd, z, 'fine_')
        rgb, occu, depth, disp, _ = self._accumulate(rgbs, z, rayd, want_weights=False)
        return pred_coarse, {'rgb': rgb, 'occu': occu, 'depth': depth, 'disp': disp}

    def _accumulate(self, rgbs, z, rayd, want_weights=True):
        noise_std = self.config.getfloat('DEFAULT', 'noise_std')
        noise = torch.randn_like(z) * noise_std if noise_std > 0 else None
        if torch.is_grad_enabled() and rgbs.requires_grad:
            return autograd.Composite.apply(rgbs, z, rayd, noise, self.white_bg, want_weights)
        return ops.composite_fwd(rgbs, z, rayd, white_bg=self.white_bg, noise=noise,
                                 want_weights=want_weights)

    def _eval_nerf_at(self, pts, views, use_fine=False):
        """rgbs[N,S,4] at explicit points (reference signature, nerf.py:256-290): expressed through
        the fused kernel as rays with origin = pts, z = 0."""
        n, s = pts.shape[:2]
        o = pts.reshape(-1, 3).contiguous()
        d = views.reshape(-1, 3).contiguous()
        z = torch.zeros((n * s, 1), dtype=torch.float32, device=pts.device)
        pref = 'fine_' if use_fine else 'coarse_'
        if not self.tuned:
            return self._eval_rays_generic(o, d, z, pref).reshape(n, s, 4)
        return ops.nerf_mlp_fwd(o, d, z, self._nerf_blob(pref), self.precision).reshape(n, s, 4)

    # ------------------------------------------------------------------ geometry extraction (geometry_from_nerf.py)
    def _sigma_generic(self, rayo, rayd, z, pref, want_normal=False):
        """(sigma_raw[N,S], normal[N,S,3] | None) of a non-shipped shape on the runtime-shaped kernels, `mlp_chunk` points at
        a time: the density through enc -> sigma_out (or column 3 of rgbs_out); the normal -l2_normalize(d relu(sigma)/dx)
        (geometry_from_nerf.py:288-297) by the input-gradient mode of nfx_mlp_generic_bwd through both networks and
        nfx_embed_bwd through the positional encoding — operands = `precision` (bf16, or the kernels' fp32 instantiation;
        the hi / lo density-gradient kernel is the shipped shape's)."""
        n, s = z.shape
        lx = self.embedder['xyz'].n_freqs
        pts = (rayo[:, None, :] + rayd[:, None, :] * z[:, :, None]).reshape(-1, 3).contiguous()
        head = pref + ('sigma_out' if self.use_views else 'rgbs_out')
        enc, out = self._generic_net(pref + 'enc', train=want_normal), self._generic_net(head, train=want_normal)
        chunk = self.config.getint('DEFAULT', 'mlp_chunk')
        sigma = torch.empty(n * s, dtype=torch.float32, device=z.device)
        normal = torch.empty((n * s, 3), dtype=torch.float32, device=z.device) if want_normal else None
        for i in range(0, n * s, chunk):
            p = pts[i:i + chunk]
            emb = ops.embed(lx, x=p)
            feat = self._enc_out(pref, ops.mlp_generic_fwd(emb, enc), emb)
            raw = ops.mlp_generic_fwd(feat, out)               # [m, 1] or [m, 4] (use_views = False: rgb + sigma)
            sigma[i:i + chunk] = raw[:, -1]
            if want_normal:
                dy = torch.zeros_like(raw)
                dy[:, -1] = (raw[:, -1] > 0).float()           # d relu(sigma) / d sigma
                d_feat = ops.mlp_generic_bwd(feat, out, dy, None, None, want_dx=True)
                d_emb = ops.mlp_generic_bwd(emb, enc, d_feat[:, :enc.d_out].contiguous(), None, None, want_dx=True)
                if d_feat.shape[1] > enc.d_out:                # (a trailing skip hands the embedding to the head directly)
                    d_emb = d_emb + d_feat[:, enc.d_out:]
                normal[i:i + chunk] = -ops.l2_normalize3(ops.embed_bwd(lx, p, d_emb), 1e-12)
        return sigma.view(n, s), (normal.view(n, s, 3) if want_normal else None)

    def _nerf_geom_blob(self, pref, precision=None):
        precision = precision or self.precision
        ks, bs = self._nerf_params(pref)
        return self._packed(pref + 'geom' + precision, ks + bs,
                            lambda k, b: ops.pack_nerf_geom_weights(k, b, precision))

    @staticmethod
    def _in_bounds(rayo, rayd, z, bbox):
        pts = rayo[:, None, :] + rayd[:, None, :] * z[:, :, None]
        lo = pts.new_tensor(bbox[0::2])
        hi = pts.new_tensor(bbox[1::2])
        return ((pts >= lo) & (pts <= hi)).all(-1)

    def eval_sigma(self, rayo, rayd, z, use_fine=False, bbox=None):
        """relu(sigma)[N,S] at rayo + rayd z (eval_sigma_mlp, geometry_from_nerf.py:322-350); outside the optional
        bounding box (x_min, x_max, y_min, y_max, z_min, z_max) the density is 0."""
        pref = 'fine_' if use_fine else 'coarse_'
        if not self.tuned:
            sigma = torch.relu(self._sigma_generic(rayo, rayd, z, pref)[0])
        else:
            raw = ops.nerf_sigma_fwd(rayo, rayd, z, self._nerf_geom_blob(pref), self.precision)
            sigma = torch.relu(self._refine_last_sigma(rayo, rayd, z, raw, pref))
        if bbox is not None:
            sigma = sigma * self._in_bounds(rayo, rayd, z, bbox)
        return sigma

    def _refine_last_sigma(self, rayo, rayd, z, sigma_raw, pref):
        """geometry_from_nerf composites these densities with accumulate_sigma, where the LAST sample of a ray gets
        dist = 1e10 (nerf.py:186-191): alpha_last = [sigma_last > 0] exactly, the one bit of a ray a bf16 density can get
        wrong by a whole ray.  As in the render (last_sample_precision), precision = bf16 re-evaluates
        that one sample with the fp32-class density kernel: 1 of 128 / 320 samples at ~3x the cost (round 5)."""
        if self.precision == 'bf16' and self.last_sample_precision == 'fp32' and z.shape[1] > 1 and z.shape[0] > 0:
            last = ops.nerf_sigma_fwd(rayo, rayd, z[:, -1:].contiguous(), self._nerf_geom_blob(pref, 'fp32'), 'fp32')
            sigma_raw[:, -1] = last[:, 0]
        return sigma_raw

    def eval_sigma_normal(self, rayo, rayd, z, bbox=None):
        """(relu(sigma)[N,S], normal[N,S,3]) of the FINE network, normal = -l2_normalize(d sigma / dx)
        (geometry_from_nerf.py:280-306: the bounding box zeroes sigma, not the normal)."""
        if not self.tuned:
            sigma, normal = self._sigma_generic(rayo, rayd, z, 'fine_', want_normal=True)
        else:
            normal, sigma = ops.nerf_sigma_grad(rayo, rayd, z, self._nerf_geom_blob('fine_'), self.precision)
            sigma = self._refine_last_sigma(rayo, rayd, z, sigma, 'fine_')
        sigma = torch.relu(sigma)
        if bbox is not None:
            sigma = sigma * self._in_bounds(rayo, rayd, z, bbox)
        return sigma, normal

    # ------------------------------------------------------------------ loss
    def compute_loss(self, pred, gt, **kwargs):
        coarse, fine = pred['coarse'], pred['fine']
        if (gt.is_cuda and fine is not None and kwargs.get('keep_batch') and kwargs.get('weights') is None
                and gt.ndim == 2 and all(isinstance(fn, losses.L2) for _, fn in self.wloss)):
            # nerf.py:292-300 per ray — sum over the loss terms of w (mse(gt, coarse) + mse(gt, fine)) — as one forward and one
            # backward launch (autograd.PairLoss, the surface models' loss kernel) instead of ~20 elementwise ones
            spec = []
            for weight, _ in self.wloss:
                spec += [(0, 1, float(weight), 'mse', False, False), (0, 2, float(weight), 'mse', False, False)]
            return nfx_grad.PairLoss.apply(None, 0., tuple(spec), gt, coarse, fine)
        loss = 0
        for weight, fn in self.wloss:
            loss = loss + weight * fn(gt, coarse, **kwargs)
            if fine is not None:
                loss = loss + weight * fn(gt, fine, **kwargs)
        return loss
