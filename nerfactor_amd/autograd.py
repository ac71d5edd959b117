"""torch.autograd glue: every differentiable hot-path op = one libnfx forward kernel + one libnfx
backward kernel.  Gradients exist for what the reference trains (trainvali.py:278-285): the MLP
kernels/biases, the light, and — between kernels — normals, albedo, roughness and visibility.
Points, cameras and light geometry are data (no gradient)."""
import os

import torch

from . import _capi, ops

# Operand type of every TUNED backward kernel (fused dgrad chains, weight-gradient GEMMs) and of the train blobs they read.
# (grad_precision = fp32 — the default of precision = fp32 — does not come here: the models route every network's
# training call through GenericMlp below on the fp32 runtime-shaped kernels, csrc/mlp_generic.hip.)
# `precision = fp32` selects the fp32-class forward kernels (bf16 hi / lo operand pairs, three MFMAs per product); its
# gradients are still formed by the bf16-operand backward kernels, which re-compute their own bf16 forward for the ReLU
# masks — mixed precision in the usual sense: fp32-class values and losses, gradients with bf16 operand rounding
# (~0.4 % relative noise, tests/test_gpu_reference_grads.py).  The C-ABI backward entry points themselves return
# NFX_ENOSUP for NFX_PREC_FP32 (include/nfx.h): there is no hi / lo backward kernel.
GRAD_PREC = 'bf16'


def _targets(params):
    """Where the backward kernels accumulate the gradient of each parameter.  The libnfx weight-gradient kernels ADD
    into their output, so when a parameter already owns a gradient buffer (optim.AMSGrad makes every .grad a view of
    its flat bucket, zeroed once per step) they add straight into it and autograd gets None for that input: no
    zero-fill and no `grad += g` kernel per parameter and call (≈100 tiny launches per NeRF step).  Otherwise a fresh
    zero buffer is returned to autograd as usual."""
    bufs, rets = [], []
    for p in params:
        g = getattr(p, 'grad', None)
        if g is not None and g.is_contiguous() and g.shape == p.shape:
            bufs.append(g)
            rets.append(None)
        else:
            z = torch.zeros_like(p)
            bufs.append(z)
            rets.append(z)
    return bufs, rets


# The backward of an xyz head (normal, albedo, BRDF code: 2048 rows of a 1024-ray step) is three launches of 16 workgroups
# each — 35 us on 6 % of the chip, latency-bound by one 128-row tile per workgroup — and a NeRFactor step has three of them.
# Nothing in the autograd graph waits for them (no input gradient; the parameter gradients are accumulated in place, _targets),
# so a head's backward is only RECORDED here and all recorded heads over the same rows leave as ONE launch pair + reduction
# (ops.mlp128_bwd_heads, blockIdx.y = head) when autograd's backward pass ends (an engine callback).  Bit-identical to one
# launch per head; heads that share a gradient buffer (the same network evaluated twice) go to separate launches.
import os as _os
BATCH_HEADS = _os.environ.get('NFX_BATCH_HEADS', '1') != '0'      # (A / B switch of the plugin, not of libnfx)
_heads = {'pending': [], 'armed': False}


def _flush_heads():
    pending, _heads['pending'], _heads['armed'] = _heads['pending'], [], False
    for h in pending:       # the callback runs on the caller's stream: wait for the streams the recorded gradients were produced on
        if h.get('event') is not None:
            torch.cuda.current_stream(h['dout'].device).wait_event(h['event'])
    while pending:
        first = pending[0]
        group, rest, bufs = [], [], set()
        for h in pending:
            same = h['xyz'] is first['xyz'] and h['xyz_scale'] == first['xyz_scale']
            ptr = h['dks'][0].data_ptr()
            if same and ptr not in bufs and len(group) < ops.MLP128_MAX_HEADS:
                group.append(h)
                bufs.add(ptr)
            else:
                rest.append(h)
        ops.mlp128_bwd_heads(_capi.IN_XYZ, first['xyz'], [(h['dout'], h['blob'], h['dks'], h['dbs'], h['out_act'], h['post_scale'])
                                                          for h in group], xyz_scale=first['xyz_scale'], prec=GRAD_PREC)
        pending = rest


class Mlp128Xyz(torch.autograd.Function):
    """out = post_scale * act(out(mlp(posenc10(xyz_scale * xyz)))) + post_bias."""

    @staticmethod
    def forward(ctx, xyz, fwd_blob, train_blob_fn, prec, out_dim, out_act, xyz_scale, post_scale, post_bias,
                *params):
        ctx.save_for_backward(xyz)
        ctx.cfg = (train_blob_fn, prec, out_dim, out_act, xyz_scale, post_scale, params)
        if (_heads['pending'] or _heads['armed']) and torch._C._current_graph_task_id() < 0:
            # heads recorded by a backward pass that never reached its end (an exception inside autograd: the engine runs no
            # callback then) must not leak into the next one.  A forward that runs WHILE a backward pass is in flight
            # (activation re-computation, a second model: the engine's graph-task id is >= 0 on this thread) leaves the
            # recorded heads alone: their gradients are still owed.
            import warnings
            warnings.warn("nerfactor_amd.autograd: dropping %d width-128 head(s) recorded by a backward pass that did not "
                          "finish" % len(_heads['pending']))
            _heads['pending'], _heads['armed'] = [], False
        return ops.mlp128_xyz_fwd(xyz, fwd_blob, out_dim, out_act=out_act, xyz_scale=xyz_scale,
                                  post_scale=post_scale, post_bias=post_bias, prec=prec)

    @staticmethod
    def backward(ctx, dout):
        (xyz,) = ctx.saved_tensors
        train_blob_fn, prec, out_dim, out_act, xyz_scale, post_scale, params = ctx.cfg
        ks, bs = list(params[:5]), list(params[5:])
        (dks, rks), (dbs, rbs) = _targets(ks), _targets(bs)
        if BATCH_HEADS and all(r is None for r in rks + rbs):      # (every gradient accumulated in place: nothing to hand back)
            dout = dout.contiguous()
            event = None
            if dout.is_cuda:    # (every gradient is returned as None: no AccumulateGrad makes the engine sync this node's stream)
                event = torch.cuda.Event()
                event.record(torch.cuda.current_stream(dout.device))
            _heads['pending'].append(dict(xyz=xyz, dout=dout, blob=train_blob_fn(), dks=dks, dbs=dbs, out_act=out_act,
                                          xyz_scale=xyz_scale, post_scale=post_scale, event=event))
            if not _heads['armed']:
                torch.autograd.Variable._execution_engine.queue_callback(_flush_heads)
                _heads['armed'] = True
        else:
            ops.mlp128_bwd(_capi.IN_XYZ, xyz, dout.contiguous(), train_blob_fn(), dks, dbs, out_act=out_act,
                           xyz_scale=xyz_scale, post_scale=post_scale, prec=GRAD_PREC)
        return (None,) * 9 + tuple(rks) + tuple(rbs)


class Lvis(torch.autograd.Function):
    """lvis[n, L] = sigmoid(out(mlp([posenc10(xyz_scale*xyz), posenc4(dir(lxyz - xyz_dir))])))."""

    @staticmethod
    def forward(ctx, xyz, xyz_dir, lxyz, fwd_blob, train_blob_fn, prec, xyz_scale, *params):
        ctx.save_for_backward(xyz, xyz_dir, lxyz)
        ctx.cfg = (train_blob_fn, prec, xyz_scale, params)
        return ops.lvis_fwd(xyz, lxyz, fwd_blob, xyz_scale=xyz_scale, xyz_dir=xyz_dir, prec=prec)

    @staticmethod
    def backward(ctx, dout):
        xyz, xyz_dir, lxyz = ctx.saved_tensors
        train_blob_fn, prec, xyz_scale, params = ctx.cfg
        ks, bs = list(params[:5]), list(params[5:])
        (dks, rks), (dbs, rbs) = _targets(ks), _targets(bs)
        ops.mlp128_bwd(_capi.IN_XYZ_LDIR, xyz, dout.contiguous(), train_blob_fn(), dks, dbs, out_act='sigmoid',
                       xyz_scale=xyz_scale, lxyz=lxyz, xyz_dir=xyz_dir, prec=GRAD_PREC)
        return (None,) * 7 + tuple(rks) + tuple(rbs)


class LvisFp32Class(torch.autograd.Function):
    """lvis[n, L] of the SHIPPED light-visibility network at grad_precision = fp32 with fp32_matrix = pairs (round 5): the
    forward is the tuned fp32-class kernel (mlp128_x3.hip: 339 TFLOP/s algorithmic against ~95 of the runtime-shaped one),
    the backward the fp32-class runtime-shaped kernel on explicit rows [posenc(xyz_scale x) | posenc(dir(light - x_dir))]
    — rebuilt chunk by chunk in the backward, so nothing of size rows x 90 lives between the two (the runtime-shaped
    forward kept 377 MB of rows per 1024-ray step for autograd).  Both evaluate the same function with 16-bit operand
    pairs; the backward re-computes its own forward for the ReLU masks."""

    @staticmethod
    def forward(ctx, xyz, xyz_dir, lxyz, fwd_blob, net_fn, xyz_scale, n_freqs, chunk_rows, *params):
        ctx.save_for_backward(xyz, xyz_dir, lxyz)
        ctx.cfg = (net_fn, xyz_scale, n_freqs, chunk_rows, params)
        return ops.lvis_fwd(xyz, lxyz, fwd_blob, xyz_scale=xyz_scale, xyz_dir=xyz_dir, prec='fp32')

    @staticmethod
    def backward(ctx, dout):
        xyz, xyz_dir, lxyz = ctx.saved_tensors
        net_fn, xyz_scale, (lx, ll), chunk_rows, params = ctx.cfg
        nk = len(params) // 2
        (dks, rks), (dbs, rbs) = _targets(params[:nk]), _targets(params[nk:])
        n, nl = xyz.shape[0], lxyz.shape[0]
        dx_cols = 3 + 6 * lx
        per = max(1, chunk_rows // nl)
        dout = dout.contiguous()
        net = net_fn()
        for i in range(0, n, per):
            x = (xyz[i:i + per] * xyz_scale).contiguous()
            rows = torch.empty((x.shape[0] * nl, dx_cols + 3 + 6 * ll), dtype=torch.float32, device=xyz.device)
            ops.embed(lx, x=x, per_ray=nl, out=rows)
            ops.embed(ll, x=xyz_dir[i:i + per].contiguous(), lights=lxyz, out=rows, col0=dx_cols)
            ops.mlp_generic_bwd(rows, net, dout[i:i + per].reshape(-1, 1), dks, dbs)
        return (None,) * 8 + tuple(rks) + tuple(rbs)


class ShadeMicrofacet(torch.autograd.Function):
    """rgb[n,3] under the trained light with the GGX microfacet BRDF (nerfactor.py:315-342)."""

    @staticmethod
    def forward(ctx, xyz, cam, lxyz, lareas, f0, to_srgb, normal, albedo, rough, lvis, light):
        ctx.save_for_backward(xyz, cam, lxyz, lareas, normal, albedo, rough, lvis, light)
        ctx.cfg = (f0, to_srgb)
        out = ops.shade_fwd(xyz, cam, normal, albedo, lvis, lxyz, lareas, light.reshape(1, -1, 3).contiguous(),
                            rough=rough, f0=f0, linear2srgb=to_srgb)
        return out[:, 0]

    @staticmethod
    def backward(ctx, drgb):
        xyz, cam, lxyz, lareas, normal, albedo, rough, lvis, light = ctx.saved_tensors
        f0, to_srgb = ctx.cfg
        d_light = torch.zeros_like(light.reshape(-1, 3))
        d_albedo, d_normal, d_lvis, d_rough = ops.shade_bwd(
            xyz, cam, normal, albedo, lvis, lxyz, lareas, light.reshape(-1, 3).contiguous(), drgb.contiguous(),
            d_light, rough=rough, f0=f0, linear2srgb=to_srgb)
        return (None,) * 6 + (d_normal, d_albedo, d_rough.reshape(rough.shape), d_lvis, d_light.reshape(light.shape))


class BrdfSpec(torch.autograd.Function):
    """spec[n, L] of the frozen learned BRDF; differentiable w.r.t. the normal and the latent z."""

    @staticmethod
    def forward(ctx, xyz, cam, lxyz, fwd_blob, train_blob_fn, prec, normal, z):
        ctx.save_for_backward(xyz, cam, lxyz, normal, z)
        ctx.train_blob_fn, ctx.prec = train_blob_fn, prec
        return ops.brdf_spec_fwd(xyz, cam, normal, z, lxyz, fwd_blob, prec=prec)

    @staticmethod
    def backward(ctx, dspec):
        xyz, cam, lxyz, normal, z = ctx.saved_tensors
        d_z, d_normal = ops.brdf_spec_bwd(xyz, cam, normal, z, lxyz, ctx.train_blob_fn(), dspec.contiguous(),
                                          prec=GRAD_PREC)
        return (None,) * 6 + (d_normal, d_z)


class BrdfRows(torch.autograd.Function):
    """(brdf, brdf_reci)[n] of the BRDF prior on explicit (z, Rusinkiewicz) rows — models/brdf.py:_eval_brdf_at.
    Differentiable w.r.t. the MLP parameters and z (the latent codes); the Rusinkiewicz coordinates are data."""

    @staticmethod
    def forward(ctx, z, rusink, train_blob_fn, prec, *params):
        ctx.save_for_backward(z, rusink)
        ctx.cfg = (train_blob_fn, prec, params)
        out = ops.brdf_rows_fwd(z, rusink, train_blob_fn(), reci=True, prec=GRAD_PREC)   # (the rows kernels: bf16 only)
        n = z.shape[0]
        return out[:n], out[n:]

    @staticmethod
    def backward(ctx, d_brdf, d_reci):
        z, rusink = ctx.saved_tensors
        train_blob_fn, prec, params = ctx.cfg
        ks, bs = list(params[:5]), list(params[5:])
        (dks, rks), (dbs, rbs) = _targets(ks), _targets(bs)
        n = z.shape[0]
        dout = torch.cat((d_brdf.reshape(n), d_reci.reshape(n)))
        d_rows = ops.brdf_rows_bwd(z, rusink, train_blob_fn(), dout, dks, dbs, reci=True, prec=GRAD_PREC)
        d_z = (d_rows[:n] + d_rows[n:]) if ctx.needs_input_grad[0] else None
        return (d_z, None, None, None) + tuple(rks) + tuple(rbs)


class ShadeSpec(torch.autograd.Function):
    """rgb[n,3] under the trained light with brdf = albedo/pi + spec_scale * spec (nerfactor.py:459-461)."""

    @staticmethod
    def forward(ctx, xyz, cam, lxyz, lareas, spec_scale, to_srgb, normal, albedo, spec, lvis, light):
        ctx.save_for_backward(xyz, cam, lxyz, lareas, normal, albedo, spec, lvis, light)
        ctx.cfg = (spec_scale, to_srgb)
        out = ops.shade_fwd(xyz, cam, normal, albedo, lvis, lxyz, lareas, light.reshape(1, -1, 3).contiguous(),
                            spec=spec, spec_scale=spec_scale, linear2srgb=to_srgb)
        return out[:, 0]

    @staticmethod
    def backward(ctx, drgb):
        xyz, cam, lxyz, lareas, normal, albedo, spec, lvis, light = ctx.saved_tensors
        spec_scale, to_srgb = ctx.cfg
        d_light = torch.zeros_like(light.reshape(-1, 3))
        d_albedo, d_normal, d_lvis, d_spec = ops.shade_bwd(
            xyz, cam, normal, albedo, lvis, lxyz, lareas, light.reshape(-1, 3).contiguous(), drgb.contiguous(),
            d_light, spec=spec, spec_scale=spec_scale, linear2srgb=to_srgb)
        return (None,) * 6 + (d_normal, d_albedo, d_spec, d_lvis, d_light.reshape(light.shape))


# 'capture' (default): the two NeRF networks' backward chains fork onto side streams while the step is being captured into a
# hipGraph (optim.GraphedTrainStep: two parallel branches of the graph; measured -2.5 ... -3 % of the replayed step when the
# chains' tile counts leave a partly filled round, profiles/r06/nerf_train_ab.txt) and stay on the caller's stream in an eager
# step (whose host side the stream switches only lengthen).  True / NFX_NERF_BWD_SIDE_STREAMS=1: always; False / =0: never.
NERF_BWD_SIDE_STREAMS = {'0': False, '1': True}.get(os.environ.get('NFX_NERF_BWD_SIDE_STREAMS', ''), 'capture')
_nerf_side = {'pending': [], 'armed': False, 'streams': {}}


def _side_stream(dev, k):
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), k % 2)
    if key not in _nerf_side['streams']:
        _nerf_side['streams'][key] = torch.cuda.Stream(device=dev)
    return _nerf_side['streams'][key]


def _join_side_streams():
    pending, _nerf_side['pending'], _nerf_side['armed'] = _nerf_side['pending'], [], False
    for event, dev, _keep in pending:
        torch.cuda.current_stream(dev).wait_event(event)


class NerfMlp(torch.autograd.Function):
    """rgbs[N,S,4] = NeRF MLP at rayo + rayd z (nerf.py:256-290); gradients for the 12 kernels / biases only."""

    @staticmethod
    def forward(ctx, rayo, rayd, z, fwd_blob, train_blob_fn, prec, *params):
        if _nerf_side['pending'] and torch._C._current_graph_task_id() < 0:
            _join_side_streams()      # (a backward pass that raised never reached its callback)
        ctx.save_for_backward(rayo, rayd, z)
        ctx.cfg = (train_blob_fn, prec, params)
        return ops.nerf_mlp_fwd(rayo, rayd, z, fwd_blob, prec)

    @staticmethod
    def backward(ctx, d_rgbs):
        rayo, rayd, z = ctx.saved_tensors
        train_blob_fn, prec, params = ctx.cfg
        ks, bs = list(params[:12]), list(params[12:])
        (dks, rks), (dbs, rbs) = _targets(ks), _targets(bs)
        d_rgbs = d_rgbs.contiguous()
        in_place = all(r is None for r in rks) and all(r is None for r in rbs)
        forked = NERF_BWD_SIDE_STREAMS is True or (NERF_BWD_SIDE_STREAMS == 'capture' and d_rgbs.is_cuda and
                                                   torch.cuda.is_current_stream_capturing())
        if forked and in_place and d_rgbs.is_cuda and torch._C._current_graph_task_id() >= 0:
            # The coarse and the fine network's backward are independent chains (nerf.py:292-300: two L2 terms, the fine
            # samples are drawn from stop-gradient weights) of kernels that each take whole CUs: a chain alone leaves the
            # CUs of its last, partly filled round idle (fine: 2.2 rounds of 256-row tiles).  Each chain goes to a side
            # stream forked from the caller's; the caller's stream joins them when the backward pass ends (the engine's
            # callback), so whoever reads .grad after loss.backward() sees finished gradients.  Only when every gradient
            # is accumulated in place: a buffer handed back to autograd would be read on the caller's stream at once.
            dev = d_rgbs.device
            cur = torch.cuda.current_stream(dev)
            side = _side_stream(dev, len(_nerf_side['pending']))
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                ws = ops.nerf_mlp_bwd(rayo, rayd, z, d_rgbs, train_blob_fn(), dks, dbs, GRAD_PREC)
                event = side.record_event()
            # (the tensors stay referenced until the join: nothing of them is handed back to the allocator of the
            #  caller's stream while the side stream still reads it)
            _nerf_side['pending'].append((event, dev, (d_rgbs, rayo, rayd, z, ws)))
            if not _nerf_side['armed']:
                torch.autograd.Variable._execution_engine.queue_callback(_join_side_streams)
                _nerf_side['armed'] = True
        else:
            ops.nerf_mlp_bwd(rayo, rayd, z, d_rgbs, train_blob_fn(), dks, dbs, GRAD_PREC)
        return (None,) * 6 + tuple(rks) + tuple(rbs)


class GenericMlp(torch.autograd.Function):
    """y = net(x) for an mlp.Network of any shape (csrc/mlp_generic.hip); gradients for its kernels / biases and — when
    x itself carries a gradient (bottleneck -> rgb_out, nerf.py:277-287) — for x.  `net_fn()` returns the
    ops.GenericNet holding the current train blob."""

    @staticmethod
    def forward(ctx, x, net_fn, *params):
        net = net_fn()
        if x.dim() != 2 or x.shape[1] != net.d_in:      # a wider x would get a gradient of the wrong shape back
            raise _capi.NfxError("GenericMlp: x must be [n, %d] (the network's input width), got %s"
                                     % (net.d_in, tuple(x.shape)))
        ctx.save_for_backward(x)
        ctx.cfg = (net_fn, params)
        return ops.mlp_generic_fwd(x, net)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        net_fn, params = ctx.cfg
        if not any(ctx.needs_input_grad[2:]):      # a frozen network (NeRFactor's BRDF prior): the input gradient only
            dx = ops.mlp_generic_bwd(x, net_fn(), dy.contiguous(), None, None, want_dx=True)
            return (dx, None) + (None,) * len(params)
        nk = len(params) // 2
        (dks, rks), (dbs, rbs) = _targets(params[:nk]), _targets(params[nk:])
        dx = ops.mlp_generic_bwd(x, net_fn(), dy.contiguous(), dks, dbs, want_dx=ctx.needs_input_grad[0])
        return (dx, None) + tuple(rks) + tuple(rbs)


class L2NormalizeRows(torch.autograd.Function):
    """tf.linalg.l2_normalize(x, axis=1, epsilon=eps) with its pull-back as ONE launch each (csrc/regularizers.hip) — under
    autograd the torch formula (util/math.py:safe_l2_normalize) is 5 launches forward and 12 backward per call."""

    @staticmethod
    def forward(ctx, x, eps):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.eps = eps
        return ops.l2_normalize_rows(x, eps)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return ops.l2_normalize_rows_bwd(x, dy.contiguous(), ctx.eps), None


def l2_normalize(v, eps=1e-6):
    """safe_l2_normalize(v, axis=1) of v[n, d]: one libnfx launch, and one more for the pull-back while autograd is
    recording; host tensors (and d > 16) take the torch formula."""
    if not v.is_cuda or v.dtype != torch.float32 or v.dim() != 2 or v.shape[1] > 16:
        return v * torch.rsqrt(torch.clamp(torch.sum(v * v, dim=1, keepdim=True), min=eps))
    if torch.is_grad_enabled() and v.requires_grad:
        return L2NormalizeRows.apply(v, eps)
    return ops.l2_normalize_rows(v.contiguous(), eps)


class LightSmoothness(torch.autograd.Function):
    """The light probe's spatial and cross-channel TV penalties (nerfactor.py:526-539) as a 0-dim loss: one launch computes
    the sum and its gradient, the backward scales the stored gradient."""

    @staticmethod
    def forward(ctx, light, tv_weight, achro_weight):
        loss, grad = ops.light_smoothness(light.contiguous(), tv_weight, achro_weight)
        ctx.save_for_backward(grad)
        return loss.reshape(())

    @staticmethod
    def backward(ctx, dloss):
        (grad,) = ctx.saved_tensors
        return grad * dloss, None, None


class BrdfRowsGeom(torch.autograd.Function):
    """(rows, front) = the learned BRDF's explicit fp32 input rows [z | embed(rusink)] per (point, light) and the front-lit
    flags (nerfactor.py:413-436); differentiable w.r.t. the normal (local frame + Rusinkiewicz angles, the reference's
    custom gradients) and the BRDF code z — one libnfx kernel each way (csrc/brdf_rows_geom.hip)."""

    @staticmethod
    def forward(ctx, xyz, cam, lxyz, n_freqs, normal, z):
        ctx.save_for_backward(xyz, cam, lxyz, normal)
        ctx.cfg = (n_freqs, z.shape[1])
        rows, front = ops.brdf_rows_geom_fwd(xyz, cam, normal, z, lxyz, n_freqs)
        ctx.mark_non_differentiable(front)
        ctx.set_materialize_grads(False)   # (no zero tensor per backward for the gradient of `front` that nobody reads)
        return rows, front

    @staticmethod
    def backward(ctx, d_rows, _unused):
        if d_rows is None:
            return (None,) * 6
        xyz, cam, lxyz, normal = ctx.saved_tensors
        n_freqs, z_dim = ctx.cfg
        d_normal, d_z = ops.brdf_rows_geom_bwd(xyz, cam, normal, z_dim, lxyz, n_freqs, d_rows.contiguous())
        return None, None, None, None, d_normal, d_z


class Embed(torch.autograd.Function):
    """embedder(x) for explicit vectors x[n, 3] (nfx_embed) with its pull-back (nfx_embed_bwd)."""

    @staticmethod
    def forward(ctx, x, n_freqs):
        x = x.contiguous()
        ctx.save_for_backward(x)
        ctx.n_freqs = n_freqs
        return ops.embed(n_freqs, x=x)

    @staticmethod
    def backward(ctx, d_out):
        (x,) = ctx.saved_tensors
        return ops.embed_bwd(ctx.n_freqs, x, d_out.contiguous()), None


class Composite(torch.autograd.Function):
    """(rgb, occu, depth, disp, weights) of nerf.py:184-254; the loss reaches the networks through rgb only, so
    that is the one differentiable output (occu / depth / disp / weights are returned detached)."""

    @staticmethod
    def forward(ctx, rgbs, z, rayd, noise, white_bg, want_weights):
        ctx.save_for_backward(rgbs, z, rayd, noise)
        ctx.white_bg = white_bg
        rgb, occu, depth, disp, w = ops.composite_fwd(rgbs, z, rayd, white_bg=white_bg, noise=noise,
                                                      want_weights=want_weights)
        ctx.mark_non_differentiable(occu, depth, disp)
        if w is None:
            w = rgb.new_empty(0)
        ctx.mark_non_differentiable(w)
        # (otherwise autograd hands the backward a zero-filled tensor for each of the four detached outputs: 8 of the 9
        #  fill launches of a NeRF training step)
        ctx.set_materialize_grads(False)
        return rgb, occu, depth, disp, w

    @staticmethod
    def backward(ctx, d_rgb, *unused):
        if d_rgb is None:
            return (None,) * 6
        rgbs, z, rayd, noise = ctx.saved_tensors
        d_rgbs = ops.composite_bwd(rgbs, z, rayd, d_rgb.contiguous(), white_bg=ctx.white_bg, noise=noise)
        return d_rgbs, None, None, None, None, None


class PairLoss(torch.autograd.Function):
    """Per-ray weighted sum of MSE / MAE terms between (optionally alpha-blended) tensor pairs — the surface models'
    compute_loss (nerfactor.py:463-541, shape.py:239-277) as one forward and one backward launch.

    spec = ((ia, ib, weight, kind, blend_a, blend_b), ...) indexes into `tensors`; a tensor may appear in several
    terms (its gradient is accumulated inside the kernel, in term order)."""

    @staticmethod
    def forward(ctx, alpha, bg, spec, *tensors):
        tensors = tuple(t.contiguous() for t in tensors)
        terms = [(tensors[ia], tensors[ib], w, kind, ba, bb) for ia, ib, w, kind, ba, bb in spec]
        ctx.save_for_backward(alpha, *tensors)
        ctx.bg, ctx.spec = bg, spec
        return ops.pair_loss_fwd(terms, alpha=alpha, bg=bg)

    @staticmethod
    def backward(ctx, dloss):
        alpha, *tensors = ctx.saved_tensors
        need = ctx.needs_input_grad[3:]
        bufs = [torch.empty_like(t) if nd else None for t, nd in zip(tensors, need)]
        written = [False] * len(tensors)
        terms, grads = [], []
        for ia, ib, w, kind, ba, bb in ctx.spec:
            if bufs[ia] is None and bufs[ib] is None:
                continue
            terms.append((tensors[ia], tensors[ib], w, kind, ba, bb))
            if ia == ib:
                raise ValueError("PairLoss: a term needs two different tensors")
            grads.append((bufs[ia], bufs[ib], written[ia], written[ib]))
            written[ia] = written[ia] or bufs[ia] is not None
            written[ib] = written[ib] or bufs[ib] is not None
        if terms:
            ops.pair_loss_bwd(terms, grads, dloss.contiguous(), alpha=alpha, bg=ctx.bg)
        # a tensor that needs a gradient but is in no term (cannot happen through the models) gets zeros
        out = [b if (b is None or wr) else torch.zeros_like(b) for b, wr in zip(bufs, written)]
        return (None, None, None) + tuple(out)
