"""datasets.mvs_shape.Dataset — surface buffers initialised from multi-view stereo instead of a NeRF (reference:
nerfactor/datasets/mvs_shape.py:28-121; `dataset = mvs_shape` in config/{shape_mvs,nerfactor_mvs}.ini).

Everything of a view lives in ONE directory, <mvs_root>/<view id>/{metadata.json, xyz.npy, normal.npy, lvis.npy,
alpha.png[, rgba.png]}; metadata.json holds `cam_loc`, `imh`, `imw` only — there is no camera matrix, so the ray
origins are the camera position for every pixel and the directions are zeros (the surface models never read `rayd`:
the view direction is camera - surface point).  The batch tuple and the foreground-ray sampling are those of
datasets/nerf_shape.py; <mvs_root>/lights.npz (read by models/shape.py:_gen_lights) replaces the generated light
sphere; `xyz_scale` (1e-3 in the shipped MVS configs) rescales the DTU-sized coordinates inside the MLP kernels."""
import glob
import json
from os.path import dirname, exists, join

import numpy as np

from .nerf_shape import Dataset as ShapeDataset


class Dataset(ShapeDataset):
    BUFFERS = (('xyz', 'xyz.npy'), ('normal', 'normal.npy'), ('lvis', 'lvis.npy'), ('alpha', 'alpha.png'))

    def _glob(self):
        root = self.config.get('DEFAULT', 'mvs_root')
        mode_str = 'val' if self.mode == 'vali' else self.mode
        pattern = '%s_000' % mode_str if self.debug else '%s_???' % mode_str
        keep, skipped = [], []
        for m in sorted(glob.glob(join(root, pattern, 'metadata.json'))):
            view = dirname(m)
            paths = {k: join(view, f) for k, f in self.BUFFERS}
            if self.mode != 'test':        # test cameras have no paired image
                paths['rgba'] = join(view, 'rgba.png')
            if all(exists(p) for p in paths.values()):
                keep.append(m)
                self.meta2buf[m] = paths
            else:
                skipped.append(self._parse_id(m))
        if skipped:
            print("datasets/mvs_shape: skipping %s (a paired buffer is missing)" % ', '.join(skipped))
        return keep

    def _rays_of(self, metadata_path):
        with open(metadata_path) as h:
            meta = json.load(h)
        cam = np.asarray(meta['cam_loc'], np.float32).reshape(3)
        rayo = np.broadcast_to(cam, (int(meta['imh']), int(meta['imw']), 3)).copy()
        return rayo, np.zeros_like(rayo)
