"""Optimizer and training step (reference: nerfactor/trainvali.py:110-127, 273-295).

  * all trainable parameters live in ONE flat fp32 buffer (the tensors the model sees are views into
    it), so do their gradients: a step is ONE all-reduce over [gradients | loss] (RCCL over xGMI; one
    process per GPU) followed by ONE fused AMSGrad kernel (nfx_amsgrad_step);
  * semantics of tf.keras.optimizers.Adam(amsgrad=True): epsilon = 1e-7 added to sqrt(vhat), bias
    correction folded into the step size, optional ExponentialDecay schedule
    lr * decay_rate ** (step / decay_steps).
"""
import torch

from . import dist as nfx_dist, ops


class AMSGrad:
    def __init__(self, params, lr, beta_1=0.9, beta_2=0.999, epsilon=1e-7, lr_decay_steps=-1,
                 lr_decay_rate=1., clipnorm=-1., clipvalue=-1.):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        if clipnorm > 0 and clipvalue > 0:
            raise ValueError("Both `clipnorm` and `clipvalue` are active -- turn one off")
        dev = self.params[0].device
        n = sum(p.numel() for p in self.params)
        self.flat = torch.empty(n, dtype=torch.float32, device=dev)
        self.bucket = nfx_dist.FlatBucket(self.params)  # [grads | loss]
        off = 0
        for p, gview in zip(self.params, self.bucket.views):
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + k].view_as(p)   # parameters become views of the flat buffer
            p.grad = gview                                # and their gradients views of the bucket
            off += k
        self.m = torch.zeros_like(self.flat)
        self.v = torch.zeros_like(self.flat)
        self.vhat = torch.zeros_like(self.flat)
        self.lr, self.beta_1, self.beta_2, self.epsilon = lr, beta_1, beta_2, epsilon
        self.lr_decay_steps, self.lr_decay_rate = lr_decay_steps, lr_decay_rate
        self.clipnorm, self.clipvalue = clipnorm, clipvalue
        self.iterations = 0

    def current_lr(self):
        if self.lr_decay_steps > 0:
            return self.lr * self.lr_decay_rate ** (self.iterations / self.lr_decay_steps)
        return self.lr

    def zero_grad(self):
        self.bucket.flat.zero_()

    def step(self, loss=0., lr_t_dev=None):
        """All-reduce [grads | loss] (sum over ranks), apply the update; returns the summed loss."""
        self.bucket.flat[-1] = loss
        _, total = self.bucket.all_reduce()
        g = self.bucket.flat[:-1]
        if self.clipvalue > 0:
            g.clamp_(-self.clipvalue, self.clipvalue)
        if self.clipnorm > 0:  # keras clips per variable
            for gv in self.bucket.views:
                nrm = gv.norm()
                gv.mul_(torch.clamp(self.clipnorm / (nrm + 1e-12), max=1.))
        if lr_t_dev is not None:
            # captured in a hipGraph: the step size comes from device memory (GraphedTrainStep refreshes it and does
            # the host-side bookkeeping — iterations, version counters — around every replay)
            ops.amsgrad_step_dev(self.flat, g, self.m, self.v, self.vhat, lr_t_dev, self.beta_1, self.beta_2,
                                 self.epsilon)
            return total.clone()
        lr = self.current_lr()
        self.iterations += 1
        ops.amsgrad_step(self.flat, g, self.m, self.v, self.vhat, lr, self.iterations, self.beta_1,
                         self.beta_2, self.epsilon)
        self.mark_updated()
        return total.clone()   # `total` is a view of the bucket's last slot, overwritten by the next step

    def mark_updated(self):
        """The update kernel wrote through raw pointers: tell torch (and the models' packed-blob caches, which key on the
        version counters) that the parameters changed."""
        torch.autograd.graph.increment_version(self.params)

    def state_dict(self):
        return {'m': self.m, 'v': self.v, 'vhat': self.vhat, 'iterations': self.iterations}

    def load_state_dict(self, sd):
        self.m.copy_(sd['m'])
        self.v.copy_(sd['v'])
        self.vhat.copy_(sd['vhat'])
        self.iterations = int(sd['iterations'])


def make_optimizer(model, config):
    """Adam(amsgrad) from the ini keys the reference reads (trainvali.py:110-127)."""
    get = lambda k, fb: config.getfloat('DEFAULT', k, fallback=fb)
    return AMSGrad(
        model.trainable_variables, lr=config.getfloat('DEFAULT', 'lr'),
        lr_decay_steps=config.getint('DEFAULT', 'lr_decay_steps', fallback=-1),
        lr_decay_rate=get('lr_decay_rate', 1.), clipnorm=get('clipnorm', -1.), clipvalue=get('clipvalue', -1.))


def train_step(model, batch, optimizer, global_bs):
    """distributed_train_step (trainvali.py:273-295): per-example loss summed / global batch size,
    backward through the libnfx kernels, one collective, one fused optimizer kernel."""
    if hasattr(model, 'flush_numerics'):
        model.flush_numerics()   # ships the previous step's check_numerics verdicts, raises for landed ones; never blocks
    optimizer.zero_grad()
    pred, gt, loss_kwargs, to_vis = model(batch, mode='train')
    loss_kwargs['keep_batch'] = True
    per_example = model.compute_loss(pred, gt, **loss_kwargs)
    weighted = per_example.sum() / global_bs   # tf.nn.compute_average_loss
    weighted.backward()
    total = optimizer.step(loss=weighted.detach())
    return total, _detached(to_vis)


def _detached(to_vis):
    """The visualisation tensors without their autograd history.  A caller that keeps `to_vis` across steps (trainvali
    does, for the first batches of an epoch) would otherwise keep the step's graph nodes alive, among them the
    parameters' AccumulateGrad nodes — created on the stream of THAT step; a later step captured in a hipGraph on
    torch's capture stream then finds them on the default stream, autograd joins the default stream into the capture
    ("AccumulateGrad node's stream does not match ..."), and ending the capture crashes (seen with
    scripts/bench_loader.py at the end of round 2)."""
    if isinstance(to_vis, dict):
        return {k: v.detach() if isinstance(v, torch.Tensor) else v for k, v in to_vis.items()}
    return to_vis


class GraphedTrainStep:
    """train_step captured once per batch shape in a hipGraph (torch.cuda.CUDAGraph) and replayed: the ~170 launches
    of a NeRFactor step become one graph launch.  What makes the step capturable: no host round trip inside it
    (asynchronous check_numerics verdicts, the dataset's foreground tag instead of torch.nonzero), every libnfx launch
    goes to torch's current stream, workspaces come from torch's (graph-private) allocator, and the AMSGrad kernel
    reads its step size from device memory.  Per replay the host copies the batch into the static input tensors,
    writes the step size, launches the graph, advances `iterations` and bumps the parameters' version counters.
    Batches that are not tagged foreground-only (dynamic shapes) fall back to train_step.

    The collective: torch's ProcessGroupNCCL records an all-reduce issued under stream capture into the graph (RCCL
    supports capture), so a step with the one all-reduce of the flat bucket in it replays like any other — exercised on
    hardware with a ONE-rank RCCL group only (tests/test_gpu_rccl.py: bit-identical to the step without a group).  With
    more than one rank the capture is therefore opt-in (`capture_collective=True`, ini key `hip_graph_collective`, or
    NFX_GRAPH_COLLECTIVE=1) and the default stays the eager step; gloo groups are never captured."""

    def __init__(self, model, optimizer, global_bs, warmup=2, capture_collective=None):
        self.model, self.opt, self.global_bs, self.warmup = model, optimizer, global_bs, warmup
        if capture_collective is None:
            import os
            cfg = getattr(model, 'config', None)
            capture_collective = os.environ.get('NFX_GRAPH_COLLECTIVE') == '1' or bool(
                cfg is not None and cfg.getboolean('DEFAULT', 'hip_graph_collective', fallback=False))
        self.capture_collective = capture_collective
        self.graphs = {}
        self.lr_t = torch.zeros(1, dtype=torch.float32, device=optimizer.flat.device)
        self.seen = {}

    @staticmethod
    def _key(batch):
        return tuple((tuple(t.shape), t.dtype) if isinstance(t, torch.Tensor) else None for t in batch)

    def _capturable(self, batch):
        from .nerfactor.datasets.nerf_shape import known_all_foreground
        if nfx_dist.world()[1] > 1 or nfx_dist.run_collectives_on_one_rank():      # the step issues an all-reduce
            import torch.distributed as dist
            if not (self.capture_collective and dist.get_backend() == 'nccl'):
                return False
        alpha = batch[5] if len(batch) > 5 else None       # NeRF batches (5 fields) have no compaction at all
        return alpha is None or known_all_foreground(alpha)

    def _eager(self, batch):
        return train_step(self.model, batch, self.opt, self.global_bs)

    def _body(self, static):
        model, opt = self.model, self.opt
        opt.zero_grad()
        pred, gt, loss_kwargs, to_vis = model(static, mode='train')
        loss_kwargs['keep_batch'] = True
        weighted = model.compute_loss(pred, gt, **loss_kwargs).sum() / self.global_bs
        weighted.backward()
        total = opt.step(loss=weighted.detach(), lr_t_dev=self.lr_t)
        pending = model.__dict__.get('_pending_numerics', [])
        model.__dict__['_pending_numerics'] = []
        flags = torch.stack([ok for _, ok in pending]) if pending else None
        return total, _detached(to_vis), [m for m, _ in pending], flags

    def __call__(self, batch):
        if not self._capturable(batch):
            return self._eager(batch)
        key = self._key(batch)
        n_seen = self.seen.get(key, 0)
        self.seen[key] = n_seen + 1
        if n_seen < self.warmup:            # host-side packers, hipFuncSetAttribute, allocator pools: warm before capture
            return self._eager(batch)
        model, opt = self.model, self.opt
        if key not in self.graphs:
            from .nerfactor.datasets.nerf_shape import known_all_foreground, mark_all_foreground
            model.flush_numerics(block=True)
            static = tuple(t.clone() if isinstance(t, torch.Tensor) else t for t in batch)
            if len(batch) > 5 and known_all_foreground(batch[5]):
                mark_all_foreground(static[5])
            self.lr_t.fill_(ops.amsgrad_step_size(opt.current_lr(), opt.iterations + 1, opt.beta_1, opt.beta_2))
            graph = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(graph):
                outs = self._body(static)
            self.graphs[key] = (graph, static, outs)
            # the capture did not execute anything: fall through and replay it for this batch
        graph, static, (total, to_vis, messages, flags) = self.graphs[key]
        for dst, src in zip(static, batch):
            if isinstance(dst, torch.Tensor):
                dst.copy_(src, non_blocking=True)
        self.lr_t.fill_(ops.amsgrad_step_size(opt.current_lr(), opt.iterations + 1, opt.beta_1, opt.beta_2))
        graph.replay()
        opt.iterations += 1
        opt.mark_updated()
        if flags is not None:               # this step's check_numerics verdicts, shipped like the eager path does
            model.__dict__.setdefault('_pending_numerics', []).extend(zip(messages, flags.clone().unbind(0)))
            model.flush_numerics()
        return total.clone(), self._live_vis(to_vis, batch)

    @staticmethod
    def _live_vis(to_vis, batch):
        """What the caller may keep: the captured `to_vis` tensors are the graph's static outputs, overwritten by every
        replay (trainvali keeps the first steps' entries until the end of the epoch), and its non-tensor fields froze
        at the capture-time batch — so tensors are cloned and `id` is taken from the live batch."""
        if not isinstance(to_vis, dict):
            return to_vis
        out = {k: v.clone() if isinstance(v, torch.Tensor) else v for k, v in to_vis.items()}
        if 'id' in out and len(batch) > 0 and not isinstance(batch[0], torch.Tensor):
            out['id'] = batch[0]
        return out
