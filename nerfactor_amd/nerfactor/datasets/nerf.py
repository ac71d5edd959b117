"""datasets.nerf.Dataset — posed RGBA views -> rays (reference: nerfactor/datasets/nerf.py:28-215).
Layout: <data_root>/{train,val,test}_???/{metadata.json, rgba.png}; metadata keys cam_transform_mat
(16 comma-separated floats, camera-to-world), cam_angle_x, imh, imw.
Batch tuple: (id_ list[N], hw int32[N,2], rayo[N,3], rayd[N,3], rgb[N,3])."""
import glob
import json
from os.path import basename, dirname, exists, join

import numpy as np
from PIL import Image

from .base import Dataset as BaseDataset


def load_rgba(path, imh=None):
    """uint8/uint16 PNG -> float32 [H,W,C] in [0,1], optionally resized to height imh (PIL bilinear; the
    reference uses cv2 through xiuminglib — data preparation, not the hot path)."""
    img = Image.open(path)
    arr = np.asarray(img)
    arr = arr.astype(np.float32) / float(np.iinfo(arr.dtype).max)
    if imh is not None and imh != arr.shape[0]:
        arr = resize(arr, imh)
    return arr


def resize(arr, new_h):
    """[H,W(,C)] float array -> height new_h, aspect preserved, bilinear per channel."""
    h, w = arr.shape[:2]
    new_w = int(round(new_h / h * w))
    chans = arr.reshape(h, w, -1)
    out = [np.asarray(Image.fromarray(chans[:, :, c].astype(np.float32), mode='F').resize(
        (new_w, new_h), Image.BILINEAR)) for c in range(chans.shape[2])]
    return np.stack(out, -1).reshape((new_h, new_w) + arr.shape[2:])


def gen_rays(to_world, angle_x, imh, imw, sps=1):
    """Pin-hole rays through the top-left corner of every (sub)pixel, camera looking down -z (reference
    datasets/nerf.py:172-193, ndc=False).  float64 [imh sps, imw sps, 3] origins and (un-normalised) directions."""
    n_x, n_y = imw * sps, imh * sps
    xs, ys = np.meshgrid(np.linspace(0, imw, n_x, endpoint=False), np.linspace(0, imh, n_y, endpoint=False))
    fl = .5 * imw / np.tan(.5 * angle_x)
    local = np.stack(((xs - .5 * imw) / fl, -(ys - .5 * imh) / fl, -np.ones_like(xs)), -1)
    rayd = local @ to_world[:3, :3].T
    rayo = np.broadcast_to(to_world[:3, 3], rayd.shape).copy()
    return rayo, rayd


class Dataset(BaseDataset):
    def __init__(self, config, mode, debug=False, always_all_rays=False, spp=1, device='cuda'):
        self.meta2img = {}
        sps = np.sqrt(spp)
        if sps != int(sps):
            raise ValueError("Samples per pixel must be a square number")
        self.sps = int(sps)
        self.always_all_rays = always_all_rays
        super().__init__(config, mode, debug=debug, device=device)

    def _get_batch_size(self):
        if self.mode == 'train':
            return self.config.getint('DEFAULT', 'n_rays_per_step')
        ret = self._load_cached(self.files[0])
        return int(np.prod(ret[-1].shape[:2]))

    def _glob(self):
        root = self.config.get('DEFAULT', 'data_root')
        mode_str = self.mode if self.mode in ('train', 'test') else 'val'
        metas = sorted(glob.glob(join(root, '%s_???' % mode_str, 'metadata.json')))
        if self.mode == 'test':
            return metas
        keep = []
        for m in metas:  # only cameras with a paired image
            img = join(dirname(m), 'rgba.png')
            if exists(img):
                keep.append(m)
                self.meta2img[m] = img
        return keep

    @staticmethod
    def _parse_id(metadata_path):
        return basename(dirname(metadata_path))

    def _read_camera(self, metadata_path):
        imh = self.config.getint('DEFAULT', 'imh')
        with open(metadata_path) as h:
            meta = json.load(h)
        imw = int(imh / meta['imh'] * meta['imw'])
        c2w = np.array([float(x) for x in meta['cam_transform_mat'].split(',')]).reshape(4, 4)
        return c2w, meta['cam_angle_x'], imh, imw

    def _process_example_precache(self, metadata_path):
        white_bg = self.config.getboolean('DEFAULT', 'white_bg')
        c2w, angle_x, imh, imw = self._read_camera(metadata_path)
        rayo, rayd = self._gen_rays(c2w, angle_x, imh, imw)
        rayo, rayd = rayo.astype(np.float32), rayd.astype(np.float32)
        id_ = self._parse_id(metadata_path)
        if self.mode == 'test':
            return id_, rayo, rayd, np.zeros((imh, imw, 3), np.float32)
        rgba = load_rgba(self.meta2img[metadata_path], imh)
        if rgba.ndim != 3 or rgba.shape[2] != 4:
            raise ValueError("Input image is not RGBA")
        rgb, alpha = rgba[:, :, :3], rgba[:, :, 3:]
        bg = 1. if white_bg else 0.
        return id_, rayo, rayd, (rgb * alpha + bg * (1. - alpha)).astype(np.float32)

    def _process_example_postcache(self, id_, rayo, rayd, rgb, rng=None, gather=None):
        hw = np.array(rgb.shape[:2], np.int32)
        rayo, rayd, rgb = self._sample_rays(rayo, rayd, rgb, rng=rng, gather=gather)
        n = rgb.shape[0]
        hw = np.tile(hw[None], (n, 1))
        return [id_] * n, hw if gather is None else gather('hw', hw, None), rayo, rayd, rgb

    def _sample_rays(self, rayo, rayd, rgb, rng=None, gather=None):
        flat = lambda a: a.reshape(-1, a.shape[-1])
        arrs = flat(rayo), flat(rayd), flat(rgb)
        if self.mode in ('vali', 'test') or self.always_all_rays:
            return arrs
        rng = self._batch_rng(0, 0) if rng is None else rng
        sel = rng.integers(0, rgb.shape[0] * rgb.shape[1], size=self.bs)
        if gather is None:
            return tuple(a[sel] for a in arrs)
        return tuple(gather(k, a, sel) for k, a in zip(('rayo', 'rayd', 'rgb'), arrs))

    def _gen_rays(self, to_world, angle_x, imh, imw):
        if self.config.getboolean('DEFAULT', 'ndc', fallback=False):
            raise NotImplementedError("ndc rays are marked untested in the reference and not supported")
        return gen_rays(to_world, angle_x, imh, imw, self.sps)
