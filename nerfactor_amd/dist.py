"""Multi-GPU plumbing: one process per GPU, `torch.distributed` (backend "nccl" = RCCL over xGMI on
ROCm; "gloo" for the CPU tests).

The reference's only parallelism is data parallelism over rays (tf.distribute.MirroredStrategy,
trainvali.py:259-330).  Here:
  * rendering shards rays (or views) across ranks, each rank writes its own slice — no data-path
    collective;
  * a training step issues ONE all-reduce on a flat fp32 bucket = [all gradients | scalar loss]
    (replaces the per-variable NCCL all-reduce of optimizer.apply_gradients, trainvali.py:285, and
    strategy.reduce(SUM, loss), trainvali.py:322).  1.09-4.77 MB: latency-bound, so one call.
"""
import os

import torch
import torch.distributed as dist


def rehearsal():
    """NFX_REHEARSAL=1: every rank of a torchrun launch uses GPU 0 and the collectives go over gloo — a functional
    check of the multi-process paths (sharded rendering, parameter broadcast, the gradient bucket) on a one-GPU box."""
    return os.environ.get('NFX_REHEARSAL') == '1'


def init_from_env(backend=None, device=None, force=False):
    """Initialise the default process group from torchrun's environment; returns (rank, world).  A single process gets no
    group (and no collective) unless `force`: then a group of ONE rank is created — on a GPU that is RCCL with one rank,
    and every collective of the training step (FlatBucket.all_reduce, broadcast_model) really goes through it."""
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            if world > 1:
                os.environ['MASTER_PORT'] = '29500'
            else:                      # one forced rank: any free port will do
                import socket
                with socket.socket() as s:
                    s.bind(('127.0.0.1', 0))
                    os.environ['MASTER_PORT'] = str(s.getsockname()[1])
        if rehearsal():
            backend, device = 'gloo', None
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        kwargs = {}
        if backend == 'nccl' and device is not None:
            kwargs['device_id'] = device
        with _stdout_to_stderr():      # (gloo announces its connections on the C-level stdout: a bench line must stay the only line there)
            dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world


class _stdout_to_stderr:
    """File descriptor 1 points at descriptor 2 inside the block (what native libraries print, not only sys.stdout)."""

    def __enter__(self):
        import sys
        sys.stdout.flush()
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        import sys
        sys.stdout.flush()
        os.dup2(self.saved, 1)
        os.close(self.saved)
        return False


def world():
    return (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)


def run_collectives_on_one_rank():
    """A process group of ONE rank on the nccl (RCCL) backend exists only because somebody asked for it
    (init_from_env(force=True), bench.py --force-group, the world_size-1 RCCL tests): then the step's collectives are
    issued for real — a sum over one rank leaves the values unchanged — instead of being skipped.  gloo groups of one
    rank (CPU tests of other things) keep the shortcut."""
    return dist.is_initialized() and dist.get_world_size() == 1 and dist.get_backend() == 'nccl'


def shard_range(n, rank=None, world_size=None):
    """Contiguous slice [lo, hi) of n rays owned by `rank` (sizes differ by at most one)."""
    if rank is None or world_size is None:
        rank, world_size = world()
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard(tensor, rank=None, world_size=None):
    lo, hi = shard_range(tensor.shape[0], rank, world_size)
    return tensor[lo:hi]


def max_over_ranks(value, device='cpu'):
    """Max of a Python float over all ranks (used for the wall-clock of a timed region)."""
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class FlatBucket:
    """One flat fp32 buffer holding every gradient plus one trailing scalar slot."""

    def __init__(self, params):
        self.params = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        n = sum(self.sizes)
        dev = self.params[0].device if self.params else 'cpu'
        self.flat = torch.zeros(n + 1, dtype=torch.float32, device=dev)
        self.views, off = [], 0
        for p, s in zip(self.params, self.sizes):
            self.views.append(self.flat[off:off + s].view_as(p))
            off += s

    def pack(self, grads, scalar):
        for v, g in zip(self.views, grads):
            if g is None:
                v.zero_()
            else:
                v.copy_(g)
        self.flat[-1] = scalar

    def all_reduce(self, stream=None):
        """Sum over ranks: ONE collective for all gradients and the loss.  `stream`: a torch.cuda.Stream to run the
        collective on instead of the current one — it waits for the work already queued on the current stream (the
        backward kernels that fill the bucket), and the current stream waits for it before anything queued later reads
        the bucket, so whatever the caller enqueues on the current stream BETWEEN this call and its first read of the
        bucket overlaps the collective.  optim.train_step has nothing to put there (the collective follows the last
        backward kernel and the optimizer kernel needs its result), so it passes None."""
        if dist.is_initialized() and (dist.get_world_size() > 1 or run_collectives_on_one_rank()):
            if stream is None or not self.flat.is_cuda:
                dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            else:
                cur = torch.cuda.current_stream(self.flat.device)
                stream.wait_stream(cur)
                with torch.cuda.stream(stream):
                    dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
                self.flat.record_stream(stream)
                cur.wait_stream(stream)
        return self.views, self.flat[-1]


def broadcast_model(model, optimizer=None, src=0):
    """Mirror rank `src`'s model onto every rank: all parameters and buffers (trainable or frozen) and, if given,
    the optimizer's flat parameter buffer and moments.  tf.distribute.MirroredStrategy creates every variable once
    and mirrors its initial value (trainvali.py:259-262); with one process per GPU each rank would otherwise keep
    its own random initialisation and the all-reduced gradient would be a sum of gradients taken at different
    points.  No-op for a single process (a forced one-rank RCCL group does run the broadcasts)."""
    if not dist.is_initialized() or (dist.get_world_size() == 1 and not run_collectives_on_one_rank()):
        return
    with torch.no_grad():
        if optimizer is not None:   # the trainable parameters are views of optimizer.flat: one call moves them all
            for t in (optimizer.flat, optimizer.m, optimizer.v, optimizer.vhat):
                dist.broadcast(t, src)
            own = {p.data_ptr() for p in optimizer.params}
        else:
            own = set()
        for t in list(model.parameters()) + list(model.buffers()):
            if t.data_ptr() not in own:
                dist.broadcast(t.data, src)
        it = torch.tensor([0 if optimizer is None else optimizer.iterations], dtype=torch.int64,
                          device=next(model.parameters()).device)
        dist.broadcast(it, src)
        if optimizer is not None:
            optimizer.iterations = int(it.item())
            torch.autograd.graph.increment_version(optimizer.params)   # packed-blob caches follow the new values


def local_device():
    """cuda:<LOCAL_RANK> (one process per GPU), made current."""
    dev = torch.device('cuda', 0 if rehearsal() else int(os.environ.get('LOCAL_RANK', 0)))
    torch.cuda.set_device(dev)
    return dev


def barrier():
    if dist.is_initialized():
        dist.barrier()


def sum_over_ranks(tensor):
    """Sum of a (device) scalar tensor over ranks, returned as a Python float."""
    t = tensor.detach().clone().reshape(1)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_cat(tensor, dst=0):
    """Ragged concatenation along dim 0 of every rank's tensor ON `dst` ONLY (ranks hold contiguous ray shards);
    other ranks get their own shard back.  Point-to-point: every other rank sends its shard to `dst`, which receives
    straight into its slice of the result — no padding, nothing lands on the ranks that would throw it away (an
    all_gather of an 800 x 800 x 512-light OLAT view would put the whole ~1 GB view on every GPU).  Off the hot path:
    validation / visualisation only."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tensor
    ws, rank = dist.get_world_size(), dist.get_rank()
    n = torch.tensor([tensor.shape[0]], dtype=torch.int64, device=tensor.device)
    sizes = [torch.zeros_like(n) for _ in range(ws)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    tensor = tensor.contiguous()
    if rank != dst:
        if sizes[rank]:
            dist.send(tensor, dst)
        return tensor
    out = torch.empty((sum(sizes),) + tuple(tensor.shape[1:]), dtype=tensor.dtype, device=tensor.device)
    off = 0
    for r, s in enumerate(sizes):
        if r == dst:
            out[off:off + s] = tensor
        elif s:
            dist.recv(out[off:off + s], r)
        off += s
    return out


def gather_objects(obj):
    """List of every rank's picklable `obj` (ids of the rays of a view)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return [obj]
    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, obj)
    return out
