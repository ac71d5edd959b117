"""datasets.base.Dataset — the loader contract of the reference (nerfactor/datasets/base.py:84-114) without
tf.data: `files`, `bs`, `build_pipeline(no_batch, no_shuffle)` returning an iterable of flat per-ray batch
tuples (host-side I/O; the tensors land on `device`).  One element = one view, as with `no_batch = True`.

Training batches are PREFETCHED like the reference's `.prefetch(AUTOTUNE)` (datasets/base.py:110-113): a producer thread
draws the rays of the next `prefetch` (ini key, default 2; 0 = off) batches and gathers them straight into page-locked
staging buffers while the GPU runs the current step; the consumer only issues the asynchronous host-to-device copies.
A 1024-ray NeRFactor batch is a 2 MB gather out of a 0.5 GB visibility buffer — ≈1.3 ms of host time against a 2.7 ms
training step.  What a batch contains is a pure function of (ini `seed`, mode, pipeline seed, position in the epoch):
reading ahead cannot change what later batches contain, every rank of a multi-process run draws the same batch (the
ranks then take disjoint shards of it, as MirroredStrategy distributes one dataset element, trainvali.py:85,100), and a
run is reproducible (the reference's tf.random draws are not)."""
import queue
import random
import threading

import numpy as np
import torch


class _StagingSlot:
    """Page-locked host buffers for one batch (allocated on first use, reused while the shapes last) and the event
    after which the device copies issued from them are complete."""

    def __init__(self):
        self.buffers, self.event = {}, None

    def take(self, key, shape, dtype):
        """A pinned numpy view to gather into (and the tensor that owns it)."""
        t = self.buffers.get(key)
        if t is None or tuple(t.shape) != tuple(shape) or t.numpy().dtype != dtype:
            t = torch.empty(tuple(shape), dtype=torch.from_numpy(np.empty(0, dtype)).dtype)
            self.buffers[key] = t.pin_memory() if torch.cuda.is_available() else t
        return self.buffers[key]

    def wait_until_free(self):
        if self.event is not None:
            self.event.synchronize()
            self.event = None

    def copies_issued(self):
        if torch.cuda.is_available():
            self.event = torch.cuda.Event()
            self.event.record()


class Dataset:
    def __init__(self, config, mode, debug=False, device='cuda'):
        if mode not in ('train', 'vali', 'test'):
            raise ValueError("Accepted dataset modes: 'train', 'vali', 'test', but input is %s" % mode)
        self.config, self.mode, self.debug, self.device = config, mode, debug, device
        self.files = self._glob()
        if not self.files:
            raise FileNotFoundError("No file to process into a dataset (mode %s)" % mode)
        self._cache = {}
        self._epochs_started = 0
        self.bs = self._get_batch_size()

    def _glob(self):
        raise NotImplementedError

    def _get_batch_size(self):
        if 'bs' not in self.config['DEFAULT']:
            raise ValueError("Specify batch size as 'bs' in the configuration file, or override "
                             "_get_batch_size()")
        return self.config.getint('DEFAULT', 'bs')

    def _process_example_precache(self, path):
        raise NotImplementedError

    def _process_example_postcache(self, *args, rng=None, gather=None):
        """`rng`: the generator of THIS batch (see _batch_rng); `gather(key, array, rows) -> array[rows]` (rows None: the
        whole array) may place the result in a page-locked staging buffer."""
        return args

    def _batch_rng(self, epoch, index):
        return np.random.default_rng([self.config.getint('DEFAULT', 'seed', fallback=0), len(self.mode),
                                      int(epoch) & 0x7fffffff, int(index)])

    def _batch_is_foreground_only(self):
        return False

    def _prefetch_depth(self):
        """Read-ahead depth.  Training batches: ini key `prefetch`, default 2 when they are bound for a GPU, 0 on a CPU
        device (there only when the ini asks for it: plain read-ahead without page-locked memory — what the CPU tests
        exercise).  Whole validation / test views: ini key `prefetch_views`, default 0 (set 1 or 2 to overlap the disk
        reads of the next view with the rendering of the current one; each view in flight is 1.4 GB of host memory at
        800 x 800 x 512 lights)."""
        if self.mode != 'train':
            return self.config.getint('DEFAULT', 'prefetch_views', fallback=0)
        on_gpu = torch.device(self.device).type == 'cuda' and torch.cuda.is_available()
        return self.config.getint('DEFAULT', 'prefetch', fallback=2 if on_gpu else 0)

    def get_n_views(self):
        return len(self.files)

    def _load_cached(self, path):
        """The pre-cache stage of one file.  Kept for the lifetime of the dataset only in training mode (`cache`, default
        true: every epoch revisits every view, as tf.data's .cache() serves the re-iterated training pipeline); a
        validation / test pipeline visits a view once — 1.4 GB of buffers per 800 x 800 x 512-light view — so only the
        most recent one is remembered (the batch-size probe of datasets/nerf.py reads view 0 before the pipeline)."""
        if self.mode != 'train' or not self.config.getboolean('DEFAULT', 'cache', fallback=True):
            last = self.__dict__.get('_last_loaded')
            if last is None or last[0] != path:
                last = self._last_loaded = (path, self._process_example_precache(path))
            return last[1]
        if path not in self._cache:
            self._cache[path] = self._process_example_precache(path)
        return self._cache[path]

    def build_pipeline(self, filter_predicate=None, seed=None, no_batch=False, no_shuffle=False):
        ds = self

        class _Pipe:
            def __iter__(self_inner):
                files = list(ds.files) if getattr(ds, 'keep_order', False) else sorted(ds.files)
                if filter_predicate is not None:
                    files = [f for f in files if filter_predicate(f)]
                if ds.mode == 'train' and not no_shuffle:
                    random.Random(seed).shuffle(files)
                # which rays a batch holds depends on (pipeline seed | how many epochs this dataset has started, position)
                epoch = seed if seed is not None else ds._epochs_started
                ds._epochs_started += 1
                depth = ds._prefetch_depth()
                if depth > 0:
                    yield from ds._prefetched(files, epoch, depth)
                    return
                for i, f in enumerate(files):
                    batch = ds._process_example_postcache(*ds._load_cached(f), rng=ds._batch_rng(epoch, i))
                    yield ds._to_device(batch)

            def take(self_inner, n):
                out = []
                for i, b in enumerate(self_inner):
                    if i >= n:
                        break
                    out.append(b)
                return out
        return _Pipe()

    def _to_device(self, batch):
        return tuple(torch.as_tensor(x).to(self.device, non_blocking=True)
                     if not isinstance(x, (str, list)) else x for x in batch)

    # ------------------------------------------------------------------ read-ahead
    def _prefetched(self, files, epoch, depth):
        """Batches of one epoch, produced `depth` ahead by a thread: training batches (a gather of n_rays_per_step rows)
        into page-locked staging slots; whole validation / test views as they come from the pre-cache stage — there the
        read-ahead overlaps the disk reads and decoding of the next view with the rendering of the current one (the
        reference's parallel .map + .prefetch, datasets/base.py:96-113)."""
        staged = self.mode == 'train'
        slots = [_StagingSlot() for _ in range(depth + 2)]   # depth queued + one being filled + one being copied from
        ready = queue.Queue(maxsize=depth)
        stop = threading.Event()
        device_index = None
        if torch.cuda.is_available() and torch.device(self.device).type == 'cuda':
            device_index = torch.device(self.device).index
            if device_index is None:
                device_index = torch.cuda.current_device()

        def put(item):
            while not stop.is_set():
                try:
                    ready.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def produce():
            try:
                if device_index is not None:
                    torch.cuda.set_device(device_index)    # (the current device is per thread)
                for i, f in enumerate(files):
                    slot = slots[i % len(slots)]
                    slot.wait_until_free()

                    def gather(key, array, rows, slot=slot):
                        shape = array.shape if rows is None else (len(rows),) + array.shape[1:]
                        # every slot gets its buffer the first time a key is seen: all page-locked allocations happen
                        # before the consumer has its first batch, none while a training step is being captured in a
                        # hipGraph (a hipHostMalloc from any thread would invalidate the capture)
                        for other in slots:
                            other.take(key, shape, array.dtype)
                        out = slot.take(key, shape, array.dtype)
                        if rows is None:
                            out.numpy()[...] = array
                        else:
                            np.take(array, rows, axis=0, out=out.numpy(), mode='clip')   # ('raise' buffers the output)
                        return out
                    batch = self._process_example_postcache(*self._load_cached(f), rng=self._batch_rng(epoch, i),
                                                            gather=gather if staged else None)
                    if not put((slot if staged else None, batch, None)):
                        return
                put((None, None, None))
            except BaseException as e:     # surfaces in the consumer, not in a dead thread
                put((None, None, e))

        thread = threading.Thread(target=produce, name='nfx-prefetch', daemon=True)
        thread.start()
        try:
            while True:
                slot, batch, err = ready.get()
                if err is not None:
                    raise err
                if batch is None:
                    return
                out = self._to_device(batch)
                if device_index is None and slot is not None:   # host "device": .to() aliased the staging buffers
                    out = tuple(x.clone() if isinstance(x, torch.Tensor) else x for x in out)
                if slot is not None:
                    slot.copies_issued()
                yield out
        finally:
            stop.set()
            thread.join()
