"""datasets.nerf_shape.Dataset — views + NeRF-derived surface buffers (reference:
nerfactor/datasets/nerf_shape.py:35-190).  Extra layout: <data_nerf_root>/<view id>/{alpha.png, xyz.npy,
normal.npy, lvis.npy}.  Batch tuple: (id_, hw, rayo, rayd, rgb, alpha[N,1], xyz[N,3], normal[N,3], lvis[N,L]);
training draws n_rays_per_step foreground rays (alpha > 0.9) of one view."""
import glob
from os.path import dirname, exists, join

import numpy as np

from .nerf import Dataset as NeRFDataset, load_rgba, resize



def mark_all_foreground(alpha):
    """Tags an alpha tensor whose rays are all known (on the host) to be foreground."""
    alpha._nfx_all_foreground = True
    return alpha


def known_all_foreground(alpha):
    return bool(getattr(alpha, '_nfx_all_foreground', False))


class Dataset(NeRFDataset):
    def __init__(self, config, mode, debug=False, always_all_rays=False, device='cuda'):
        self.meta2buf = {}
        self._candidates = {}   # view id -> indices of its foreground rays
        super().__init__(config, mode, debug=debug, always_all_rays=always_all_rays, device=device)

    def _glob(self):
        root = self.config.get('DEFAULT', 'data_root')
        nerf_root = self.config.get('DEFAULT', 'data_nerf_root')
        mode_str = 'val' if self.mode == 'vali' else self.mode
        pattern = '%s_002' % mode_str if self.debug else '%s_???' % mode_str
        keep = []
        for m in sorted(glob.glob(join(root, pattern, 'metadata.json'))):
            id_ = self._parse_id(m)
            paths = {k: join(nerf_root, id_, f) for k, f in
                     (('xyz', 'xyz.npy'), ('normal', 'normal.npy'), ('lvis', 'lvis.npy'), ('alpha', 'alpha.png'))}
            if self.mode != 'test':
                paths['rgba'] = join(dirname(m), 'rgba.png')
            if all(exists(p) for p in paths.values()):
                keep.append(m)
                self.meta2buf[m] = paths
        return keep

    def _rays_of(self, metadata_path):
        """(rayo, rayd) [H, W, 3] float32 of one view; the MVS variant (datasets/mvs_shape.py) has camera positions only."""
        c2w, angle_x, imh, imw = self._read_camera(metadata_path)
        rayo, rayd = self._gen_rays(c2w, angle_x, imh, imw)
        return rayo.astype(np.float32), rayd.astype(np.float32)

    def _process_example_precache(self, metadata_path):
        cfg = self.config
        imh = cfg.getint('DEFAULT', 'imh')
        rayo, rayd = self._rays_of(metadata_path)
        paths = self.meta2buf[metadata_path]
        xyz = np.load(paths['xyz']).astype(np.float32)
        normal = np.load(paths['normal']).astype(np.float32)
        if self.debug:  # fake visibility for faster debugging, as the reference does
            lvis = 0.5 * np.ones(normal.shape[:2] + (512,), np.float32)
        else:
            lvis = np.load(paths['lvis']).astype(np.float32)
        if self.mode == 'test':
            alpha = load_rgba(paths['alpha'])
            rgb = np.zeros_like(xyz)
        else:
            rgba = load_rgba(paths['rgba'])
            if rgba.ndim != 3 or rgba.shape[2] != 4:
                raise ValueError("Input image is not RGBA")
            rgb = rgba[:, :, :3]
            alpha = load_rgba(paths['alpha']) if cfg.getboolean('DEFAULT', 'use_nerf_alpha', fallback=False) else rgba[:, :, 3]
        if alpha.ndim == 3:
            alpha = alpha[:, :, 0]
        if imh != xyz.shape[0]:
            xyz, normal, lvis, alpha, rgb = (resize(a, imh) for a in (xyz, normal, lvis, alpha, rgb))
        if rayo.shape[0] != xyz.shape[0]:       # rays at the metadata's resolution, buffers at imh (MVS views)
            rayo, rayd = resize(rayo, xyz.shape[0]), resize(rayd, xyz.shape[0])
        if np.isclose(xyz, rayo).all(axis=2).any():
            raise ValueError("Found XYZs coinciding with the camera")
        normal = normal / np.maximum(np.linalg.norm(normal, axis=2, keepdims=True), 1e-12)
        lvis = np.clip(lvis, 0, 1)
        return (self._parse_id(metadata_path), rayo, rayd, rgb.astype(np.float32), alpha.astype(np.float32),
                xyz, normal.astype(np.float32), lvis)

    ALPHA_THRES = 0.9    # training rays are drawn from alpha > 0.9 (nerf_shape.py:102-107)

    def _process_example_postcache(self, id_, rayo, rayd, rgb, alpha, xyz, normal, lvis, rng=None, gather=None):
        hw = np.array(rgb.shape[:2], np.int32)
        arrs = self._sample_rays(id_, rayo, rayd, rgb, alpha, xyz, normal, lvis, rng=rng, gather=gather)
        n = arrs[2].shape[0]
        hw = np.tile(hw[None], (n, 1))
        return ([id_] * n, hw if gather is None else gather('hw', hw, None)) + arrs

    def _sample_rays(self, id_, rayo, rayd, rgb, alpha, xyz, normal, lvis, rng=None, gather=None):
        flat = lambda a: a.reshape(a.shape[0] * a.shape[1], -1)
        arrs = tuple(flat(a) for a in (rayo, rayd, rgb, alpha, xyz, normal, lvis))
        if not self._batch_is_foreground_only():
            return arrs
        cand = self._candidates.get(id_)
        if cand is None:
            cand = np.nonzero(arrs[3][:, 0] > self.ALPHA_THRES)[0]
            if self.config.getboolean('DEFAULT', 'cache', fallback=True):
                self._candidates[id_] = cand
        rng = self._batch_rng(0, 0) if rng is None else rng
        sel = cand[rng.integers(0, cand.shape[0], size=self.bs)]
        if gather is None:
            return tuple(a[sel] for a in arrs)
        return tuple(gather(k, a, sel) for k, a in zip(('rayo', 'rayd', 'rgb', 'alpha', 'xyz', 'normal', 'lvis'), arrs))

    def _batch_is_foreground_only(self):
        return self.mode == 'train' and not self.always_all_rays

    def _to_device(self, batch):
        """A training batch says that its rays are all foreground, and the models skip the foreground compaction — a
        torch.nonzero whose row count the host would have to wait for, with the whole previous step still in the
        launch queue — for it (models/nerfactor.py, models/shape.py)."""
        out = super()._to_device(batch)
        if self._batch_is_foreground_only():
            mark_all_foreground(out[5])
        return out
