"""models.base.Model — the plugin contract of the reference (nerfactor/models/base.py:25-143) on
torch.nn.Module: `config` (ConfigParser), `net` dict of networks, `register_trainable()`,
`call(batch, mode)` -> (pred, gt, loss_kwargs, to_vis), `compute_loss`, `vis_batch`,
`compile_batch_vis`, `trainable_variables`.
"""
import numpy as np
import torch

from nerfactor_amd import ops

from .. import losses
from ..networks import base as basenet



def _lights_major(a):
    """uint8 [rays, lights, 3] -> contiguous numpy [lights, rays, 3]: on the device if the rows still live there,
    otherwise on the host with each rgb triple moved as one 3-byte element (2.5x faster than a byte-wise transpose)."""
    if isinstance(a, torch.Tensor) and a.is_cuda:
        return a.permute(1, 0, 2).contiguous().cpu().numpy()
    a = np.ascontiguousarray(a.numpy() if isinstance(a, torch.Tensor) else a)
    n, nl, c = a.shape
    triples = a.view(np.dtype((np.void, c))).reshape(n, nl)
    return np.ascontiguousarray(triples.T).view(a.dtype).reshape(nl, n, c)


_PNG_POOL = None


def _png_pool():
    """One process-wide pool of PNG encoder threads (at most 32, leaving two cores to the launch threads)."""
    global _PNG_POOL
    if _PNG_POOL is None:
        import os
        from concurrent.futures import ThreadPoolExecutor
        _PNG_POOL = ThreadPoolExecutor(max_workers=max(1, min(32, (os.cpu_count() or 4) - 2)),
                                       thread_name_prefix='nfx-png')
    return _PNG_POOL


class Model(torch.nn.Module):
    DEFAULT_FP32_MATRIX = 'pairs'

    def __init__(self, config, debug=False):
        super().__init__()
        self.config = config
        self.debug = debug
        self.net = {'main': basenet.Network()}
        self.trainable_registered = False
        self.wloss = self._init_loss()
        self.precision = config.get('DEFAULT', 'precision', fallback='bf16')  # MFMA operand type
        # Operand type of the BACKWARD.  precision = fp32 trains at the reference's own arithmetic (trainvali.py:273-285
        # differentiates in fp32) by default: forward and backward of every network through the fp32 instantiation of
        # the runtime-shaped kernels (csrc/mlp_generic.hip: fp32 operands, native fp32 matrix instruction) — about 15x
        # the bf16 step.  grad_precision = bf16 keeps round 3's mixed mode (fp32-class forward kernels, bf16-operand
        # backward kernels), which is what precision = bf16 always uses.
        self.grad_precision = config.get('DEFAULT', 'grad_precision', fallback=self.precision)
        if self.grad_precision not in ('bf16', 'fp32') or (self.precision == 'bf16' and self.grad_precision == 'fp32'):
            raise ValueError("grad_precision = %s with precision = %s (bf16 | fp32; fp32 gradients need precision = fp32)"
                             % (self.grad_precision, self.precision))
        # Which fp32 instantiation of the runtime-shaped kernels `precision = fp32` runs (csrc/mlp_generic.hip): `pairs`
        # (round 5) = fp32 activations / gradients with bf16 hi / lo operand pairs on the bf16 matrix pipe — the arithmetic
        # class of the tuned fp32 render kernels, 1.7-1.9x the step rate of `native` = fp32 operands on
        # v_mfma_f32_32x32x2_f32.  The default is per model (DEFAULT_FP32_MATRIX): `pairs` where its gradients stay within
        # the 1e-3 of the reference's that tests/reference_steps.py:FP32_TOL states (the surface models and the BRDF prior:
        # <= 4.9e-4 measured), `native` for NeRF, whose 8 x 256 network and inverse-CDF sampler turn pre-activations 1e-6
        # off into 1.8e-2 on its worst gradient tensor (ReLU masks and sample bins that hang on the last bits).
        self.fp32_matrix = config.get('DEFAULT', 'fp32_matrix', fallback=self.DEFAULT_FP32_MATRIX)
        if self.fp32_matrix not in ('pairs', 'native'):
            raise ValueError("fp32_matrix = %s (pairs | native)" % self.fp32_matrix)
        self.generic_prec = 'bf16' if self.precision == 'bf16' else ('fp32' if self.fp32_matrix == 'pairs' else 'fp32_native')
        self._blobs = {}  # packed-weight cache: key -> (versions, device blob)
        self._packers = {}  # key -> ops.DevicePacker

    # ------------------------------------------------------------------ loss string parsing
    def _init_loss(self):
        wloss = []
        for item in self.config.get('DEFAULT', 'loss').split(','):
            name, weight = self._parse_loss_and_weight(item)
            if name == 'l1':
                fn = losses.L1()
            elif name == 'l2':
                fn = losses.L2()
            else:  # the reference names lpips/elpips/ssim too; no shipped config uses them
                raise NotImplementedError(name)
            wloss.append((weight, fn))
        return wloss

    @staticmethod
    def _parse_loss_and_weight(weight_loss_str):
        """'1e+2l2' -> ('l2', 100.); 'l2' -> ('l2', 1.): longest float-parsable prefix wins."""
        s = weight_loss_str.strip()
        for cut in range(len(s), 0, -1):
            try:
                return s[cut:], float(s[:cut])
            except ValueError:
                pass
        return s, 1.

    # ------------------------------------------------------------------ trainable bookkeeping
    def register_trainable(self):
        """Expose every built layer of every network in `self.net` as a direct submodule named
        net_<netname>_layer<i> — the names the reference's checkpoints are keyed on
        (models/base.py:81-104)."""
        for net_name, net in self.net.items():
            attr = 'net_' + net_name
            if not attr.isidentifier():
                raise ValueError("network name %r does not make a valid attribute" % net_name)
            for i, layer in enumerate(net.layers):
                full = '%s_layer%d' % (attr, i)
                if hasattr(self, full):
                    if getattr(self, full) is layer:
                        continue
                    raise ValueError("Can't register `%s`: attribute exists" % full)
                self.add_module(full, layer)
        self.trainable_registered = True

    @property
    def trainable_variables(self):
        return [p for p in self.parameters() if p.requires_grad]

    @staticmethod
    def _validate_mode(mode):
        if mode not in ('train', 'vali', 'test'):
            raise ValueError(mode)

    # ------------------------------------------------------------------ numerics checks
    def check_numerics(self, tensor, message):
        """tf.debugging.check_numerics: raises FloatingPointError(message) on Inf / NaN.  While autograd is recording
        (a training step) the verdict stays on the device and is raised by flush_numerics() — called by
        optim.train_step at the start of the following steps and, blocking, by the drivers at the end of an epoch — so
        that a step enqueues all of its kernels without a host round trip."""
        if tensor.is_cuda and tensor.dtype == torch.float32 and tensor.is_contiguous() and tensor.data_ptr() % 16 == 0:
            ok = ops.all_finite(tensor.detach())      # one pass, no full-size temporaries
        else:
            ok = torch.isfinite(tensor).all()
        self.check_flag(ok, message)
        return tensor

    def check_flag(self, ok, message):
        """`ok`: a 0-dim / 1-element bool tensor a kernel's own epilogue produced (round 6: nfx_lvis_fwd_rows ORs a NaN flag
        while it stores) — the verdict of check_numerics without the extra pass over the tensor."""
        ok = ok.reshape(())
        if torch.is_grad_enabled():
            self.__dict__.setdefault('_pending_numerics', []).append((message, ok))
        elif not bool(ok):
            raise FloatingPointError(message)

    def flush_numerics(self, block=False):
        """Ships the verdicts recorded since the last call to pinned host memory (asynchronous copy + event) and raises
        for every earlier group whose copy has landed.  Never waits for the GPU unless `block`: a blocking read of the
        previous step's flags would make the host wait for that whole step before issuing the next one (measured:
        1.8 ms of a 4.6 ms NeRFactor step with the GPU idle meanwhile); a NaN is reported one or two steps late."""
        import collections
        pending = self.__dict__.get('_pending_numerics', [])
        self.__dict__['_pending_numerics'] = []
        inflight = self.__dict__.setdefault('_numerics_inflight', collections.deque())
        if pending:
            flags = torch.stack([ok for _, ok in pending])
            if flags.is_cuda:
                host = torch.empty(flags.shape, dtype=flags.dtype, pin_memory=True)
                host.copy_(flags, non_blocking=True)
                event = torch.cuda.Event()
                event.record()
            else:
                host, event = flags, None
            inflight.append(([m for m, _ in pending], host, event))
        while inflight and (block or inflight[0][2] is None or inflight[0][2].query()):
            messages, host, event = inflight.popleft()
            if event is not None:
                event.synchronize()
            for message, ok in zip(messages, host):
                if not bool(ok):
                    inflight.clear()
                    raise FloatingPointError(message)

    # ------------------------------------------------------------------ packed-weight cache
    def _packed(self, key, tensors, pack_fn):
        """Device blob for `tensors` (Keras-layout parameters: kernels then biases, as many of each), re-packed only
        when one of them changed (in-place edit, checkpoint restore, optimizer step — optim.AMSGrad bumps the version
        counters, its kernel writes through raw pointers).  `pack_fn(kernels, biases)` is the host packer; after the
        first call the re-pack runs on the device (ops.DevicePacker), so a training step never leaves the GPU."""
        versions = tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self._blobs.get(key)
        dev = tensors[0].device
        if hit is not None and hit[0] == versions and hit[1].device == dev:
            return hit[1]
        if dev.type != 'cuda':
            raise RuntimeError("libnfx blobs live on the GPU: move the model with .to('cuda') first")
        if key not in self._packers:
            nk = len(tensors) // 2
            try:
                self._packers[key] = ops.DevicePacker(pack_fn, [t.shape for t in tensors[:nk]],
                                                      [t.shape for t in tensors[nk:]])
            except ops.NotAGather:
                self._packers[key] = None   # e.g. the split hi/lo fragments of precision = fp32: packed on the host
        packer = self._packers[key]
        if packer is None:
            nk = len(tensors) // 2
            blob = pack_fn(list(tensors[:nk]), list(tensors[nk:])).to(dev)
        else:
            blob = packer.pack(list(tensors))
        self._blobs[key] = (versions, blob)
        return blob

    # ------------------------------------------------------------------ contract
    def forward(self, batch, mode='train', **kwargs):
        return self.call(batch, mode=mode, **kwargs)

    def call(self, batch, mode='train'):
        raise NotImplementedError

    def compute_loss(self, pred, gt, **kwargs):
        raise NotImplementedError

    def vis_batch(self, data_dict, outdir, mode='train', dump_raw_to=None, **kwargs):
        """Writes every per-ray buffer of one full view as <key>.png (linear values clipped to [0,1]; normals
        mapped from [-1,1]; light visibility averaged over lights) plus metadata.json {"id": view}.  The
        reference's collages / videos / HTML (nerf.py:343-420, shape.py:279-360, nerfactor.py:460-640) are
        viewer tooling outside the hot path; `dump_raw_to` gets the raw tensors (np.savez instead of pickle).
        = write_vis(vis_rows(data_dict)): the two halves are separate so that N ranks can each quantise their ray
        shard and only uint8 rows travel to rank 0."""
        import os
        self._validate_mode(mode)
        if dump_raw_to is not None:
            arrays = {}
            for k, v in data_dict.items():
                if isinstance(v, torch.Tensor):
                    arrays[k] = v.detach().float().cpu().numpy()
                elif isinstance(v, (list, tuple)) and v and isinstance(v[0], str):
                    arrays[k] = np.array(v)
                elif v is not None:
                    arrays[k] = np.asarray(v)
            os.makedirs(os.path.dirname(dump_raw_to) or '.', exist_ok=True)
            with open(dump_raw_to, 'wb') as h:
                np.savez(h, **arrays)
        if mode == 'train':
            return  # random rays of one view: nothing to lay out as an image
        self.write_vis(self.vis_rows(data_dict), outdir)

    @staticmethod
    def vis_rows(data_dict):
        """Per-ray uint8 rows of every image-like buffer of `data_dict` ([n, 1 | 3] or [n, k, 3] for k lights /
        probes), plus 'id' (str) and 'hw' (int pair).  Pure per-ray arithmetic: valid on any ray shard."""
        rows = {}
        for k, v in data_dict.items():
            if k == 'id':
                rows['id'] = str(v[0]) if len(v) else ''
                continue
            if k == 'hw':
                hw = v[0] if len(v) else (0, 0)
                rows['hw'] = (int(hw[0]), int(hw[1]))
                continue
            if not isinstance(v, torch.Tensor) or v.dim() < 1 or not (v.dtype.is_floating_point or v.dtype in (
                    torch.int32, torch.int64, torch.uint8)):
                continue
            a = v.detach().float()
            if a.dim() == 3 and a.shape[2] == 3:      # [rays, lights or probes, 3]
                rows[k] = (a.clamp(0, 1) * 255 + 0.5).to(torch.uint8)
                continue
            a = a.reshape(a.shape[0], -1)
            if a.shape[1] not in (1, 3):
                a = a.mean(1, keepdim=True)           # e.g. per-light visibility
            if 'normal' in k:
                a = a / 2 + 0.5
            elif 'albedo' in k:
                a = a.clamp(0, 1) ** (1 / 2.2)        # display gamma, undone by test.py's compute_rgb_scales
            rows[k] = (a.clamp(0, 1) * 255 + 0.5).to(torch.uint8)
        return rows

    @staticmethod
    def write_vis(rows, outdir):
        """PNG files of one whole view from its uint8 rows (vis_rows of the full view, or the rank-0 concatenation
        of every rank's shard)."""
        import json
        import os
        from PIL import Image
        os.makedirs(outdir, exist_ok=True)
        h, w = rows['hw']
        with open(os.path.join(outdir, 'metadata.json'), 'w') as fh:
            json.dump({'id': rows['id']}, fh)
        jobs = []
        for k, a in rows.items():
            if k in ('id', 'hw'):
                continue
            if a.shape[0] != h * w:
                continue
            if a.ndim == 3:
                # [rays, lights, 3] -> one contiguous image per light with ONE transposing pass (on the device when the
                # rows still live there); slicing a light's column out of the ray-major buffer instead re-reads the whole
                # buffer per light — 512 passes over 1 GB for an 800 x 800 OLAT view
                a = _lights_major(a)
                os.makedirs(os.path.join(outdir, k), exist_ok=True)
                jobs += [(a[i], os.path.join(outdir, k, '%04d.png' % i)) for i in range(a.shape[0])]
                continue
            a = a.cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
            jobs.append((a, os.path.join(outdir, k + '.png')))
        # PNG encoding is the slow part of a rendered view (≈40 ms per 800 x 800 image against a 21 ms render; an OLAT
        # pass writes 512 of them): zlib releases the GIL, so the images of one view are encoded on the host's cores
        # side by side; the call still returns with every file written.
        def save(job):
            img = np.ascontiguousarray(job[0]).reshape(h, w, -1)
            Image.fromarray(img[:, :, 0] if img.shape[2] == 1 else img).save(job[1])
        if len(jobs) <= 2:
            for job in jobs:
                save(job)
        else:
            list(_png_pool().map(save, jobs))

    def compile_batch_vis(self, batch_vis_dirs, outpref, mode='train', **kwargs):
        """Index of the per-batch directories (the reference emits HTML / MP4 here)."""
        import os
        if not batch_vis_dirs:
            return None
        path = outpref + '.txt'
        os.makedirs(os.path.dirname(path) or '.', exist_ok=True)
        with open(path, 'w') as h:
            h.write('\n'.join(batch_vis_dirs) + '\n')
        return path
