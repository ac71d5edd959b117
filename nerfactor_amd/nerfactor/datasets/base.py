"""datasets.base.Dataset — the loader contract of the reference (nerfactor/datasets/base.py:84-114) without
tf.data: `files`, `bs`, `build_pipeline(no_batch, no_shuffle)` returning an iterable of flat per-ray batch
tuples (host-side I/O; the tensors land on `device`).  One element = one view, as with `no_batch = True`."""
import random

import torch


class Dataset:
    def __init__(self, config, mode, debug=False, device='cuda'):
        if mode not in ('train', 'vali', 'test'):
            raise ValueError("Accepted dataset modes: 'train', 'vali', 'test', but input is %s" % mode)
        self.config, self.mode, self.debug, self.device = config, mode, debug, device
        self.files = self._glob()
        if not self.files:
            raise FileNotFoundError("No file to process into a dataset (mode %s)" % mode)
        self._cache = {}
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

    def _process_example_postcache(self, *args):
        return args

    def get_n_views(self):
        return len(self.files)

    def _load_cached(self, path):
        if not self.config.getboolean('DEFAULT', 'cache', fallback=True):
            return self._process_example_precache(path)
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
                for f in files:
                    yield ds._to_device(ds._process_example_postcache(*ds._load_cached(f)))

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
