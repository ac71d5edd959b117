"""datasets.brdf_merl.Dataset — MERL BRDF tables for the prior (reference: nerfactor/datasets/brdf_merl.py:28-148).
Layout: <data_root>/{train,vali}_<material>.npz with arrays name, i, envmap_h, ims, spp, rusink[M,3], refl[M,1];
one shared <data_root>/test*.npz with the test Rusinkiewicz coordinates.  A batch = n_rays_per_step random rows of one
material (train) or all rows (vali / test); test ids are the seen materials followed by interpolated identities
'<n>_<w1>_<mat1>_<w2>_<mat2>'.  Batch tuple: (id_ list[N], i int32[N], envmap_h[N], ims[N], spp[N], rusink, refl)."""
import glob
from os.path import basename, join

import numpy as np

from .base import Dataset as BaseDataset


class Dataset(BaseDataset):
    keep_order = True   # test ids: seen materials first, then the interpolation sequence

    def __init__(self, config, mode, debug=False, seed=0, n_iden=20, n_between=11, device='cuda'):
        root = config.get('DEFAULT', 'data_root')
        train_paths = sorted(glob.glob(join(root, 'train_*.npz')))
        vali_paths = sorted(glob.glob(join(root, 'vali_*.npz')))
        test_paths = sorted(glob.glob(join(root, 'test*.npz')))
        self.brdf_names = [basename(x)[len('train_'):-len('.npz')] for x in train_paths]
        self.test_data = None
        test_ids = []
        if mode == 'test':
            if len(test_paths) != 1:
                raise ValueError("There should be a single set of test coordinates, shared by all identities")
            self.test_data = dict(np.load(test_paths[0], allow_pickle=True))
            test_ids += self.brdf_names                           # novel coordinates, seen identities
            rng = np.random.RandomState(seed)                     # then interpolated identities
            mats = rng.choice(self.brdf_names, min(n_iden, len(self.brdf_names)), replace=False)
            k = 0
            for a_i in range(len(mats) - 1):
                for a in np.linspace(1, 0, n_between, endpoint=True):
                    test_ids.append('%06d_%f_%s_%f_%s' % (k, a, mats[a_i], 1 - a, mats[a_i + 1]))
                    k += 1
        self.paths = {'train': train_paths, 'vali': vali_paths, 'test': test_ids}
        super().__init__(config, mode, debug=debug, device=device)

    def _get_batch_size(self):
        return self.config.getint('DEFAULT', 'n_rays_per_step')

    def get_n_brdfs(self):
        return len(self.paths[self.mode])

    def _glob(self):
        return list(self.paths[self.mode])

    def _process_example_precache(self, path):
        if self.mode == 'test':
            data, id_ = self.test_data, path
        else:
            data = np.load(path, allow_pickle=True)
        rusink = np.asarray(data['rusink'], np.float32)
        envmap_h, ims, spp = (int(np.asarray(data[k])[()]) for k in ('envmap_h', 'ims', 'spp'))
        if self.mode == 'test':
            i = self.brdf_names.index(id_) if id_ in self.brdf_names else -1
            refl = np.zeros((rusink.shape[0], 1), np.float32)    # placeholder
        else:
            id_ = str(np.asarray(data['name'])[()])
            i = int(np.asarray(data['i'])[()])
            refl = np.asarray(data['refl'], np.float32).reshape(-1, 1)
        return id_, i, envmap_h, ims, spp, rusink, refl

    def _process_example_postcache(self, id_, i, envmap_h, ims, spp, rusink, refl, rng=None, gather=None):
        if self.mode == 'train':
            rng = self._batch_rng(0, 0) if rng is None else rng
            sel = rng.integers(0, rusink.shape[0], size=self.bs)
            if gather is None:
                rusink, refl = rusink[sel], refl[sel]
            else:
                rusink, refl = gather('rusink', rusink, sel), gather('refl', refl, sel)
        n = rusink.shape[0]
        tile = lambda v: np.full((n,), v, np.int32)
        ints = np.stack([tile(v) for v in (i, envmap_h, ims, spp)])
        if gather is not None:
            ints = gather('ints', ints, None)
        return [id_] * n, ints[0], ints[1], ints[2], ints[3], rusink, refl
