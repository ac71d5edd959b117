"""Writes a tiny on-disk scene in the reference's data layout (SURVEY.md §2 datasets rows): a unit sphere seen
from cameras on a radius-4 orbit.  <root>/data/{train,val,test}_%03d/{metadata.json, rgba.png[, albedo.png]} and
<root>/nerf/<view id>/{alpha.png, xyz.npy, normal.npy, lvis.npy} (what geometry_from_nerf would have produced)."""
import json
import os
from os.path import join

import numpy as np
from PIL import Image

from oracle import nerf_ref, nerfactor_ref

ANGLE_X = 0.6911


def _view(cam_loc, imh, imw, light_xyz):
    c2w = nerf_ref.lookat_cam_to_world(cam_loc)
    rayo, rayd = nerf_ref.gen_rays(c2w, ANGLE_X, imh, imw)
    d = rayd / np.linalg.norm(rayd, axis=-1, keepdims=True)
    b = (rayo * d).sum(-1)
    disc = b * b - ((rayo * rayo).sum(-1) - 1.)
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.))
    xyz = np.where(hit[..., None], rayo + t[..., None] * d, 0.)
    normal = np.where(hit[..., None], xyz, np.array([0., 0., 1.]))
    ldir = light_xyz[None, None] - xyz[:, :, None]
    ldir /= np.linalg.norm(ldir, axis=-1, keepdims=True)
    cos = (ldir * normal[:, :, None]).sum(-1)
    lvis = np.where(hit[..., None], (cos > 0).astype(np.float64), 0.)
    albedo = 0.5 + 0.4 * np.sin(3. * xyz)            # smooth spatially-varying colour
    shade = np.clip(cos, 0, None).mean(-1, keepdims=True) * 2.
    rgb = np.clip(albedo * shade, 0, 1) * hit[..., None]
    alpha = hit.astype(np.float64)
    return c2w, rgb, albedo * hit[..., None], alpha, xyz, normal, lvis


def _png(path, arr):
    Image.fromarray((np.clip(arr, 0, 1) * 255 + .5).astype(np.uint8)).save(path)


def write_scene(root, imh=16, imw=16, n_train=3, n_val=1, n_test=2, light_h=16, seed=0):
    rng = np.random.default_rng(seed)
    light_xyz, _ = nerfactor_ref.gen_light_xyz(light_h, 2 * light_h)
    light_xyz = light_xyz.reshape(-1, 3)
    data_root, nerf_root = join(root, 'data'), join(root, 'nerf')
    for split, n in (('train', n_train), ('val', n_val), ('test', n_test)):
        for i in range(n):
            v = rng.normal(size=3)
            v[2] = abs(v[2]) + 0.3
            cam_loc = 4. * v / np.linalg.norm(v)
            c2w, rgb, albedo, alpha, xyz, normal, lvis = _view(cam_loc, imh, imw, light_xyz)
            id_ = '%s_%03d' % (split, i)
            os.makedirs(join(data_root, id_), exist_ok=True)
            os.makedirs(join(nerf_root, id_), exist_ok=True)
            with open(join(data_root, id_, 'metadata.json'), 'w') as h:
                json.dump({'cam_transform_mat': ','.join(repr(float(x)) for x in c2w.ravel()),
                           'cam_angle_x': ANGLE_X, 'imh': imh, 'imw': imw}, h)
            if split != 'test':
                _png(join(data_root, id_, 'rgba.png'), np.concatenate((rgb, alpha[..., None]), -1))
                _png(join(data_root, id_, 'albedo.png'), np.concatenate((albedo, alpha[..., None]), -1))
            _png(join(nerf_root, id_, 'alpha.png'), alpha)
            np.save(join(nerf_root, id_, 'xyz.npy'), xyz.astype(np.float32))
            np.save(join(nerf_root, id_, 'normal.npy'), normal.astype(np.float32))
            np.save(join(nerf_root, id_, 'lvis.npy'), lvis.astype(np.float32))
    return data_root, nerf_root


def write_mvs_scene(root, imh=12, imw=16, n_train=2, n_val=1, n_test=1, light_h=16, seed=3, xyz_units=1000.):
    """The same sphere in the layout datasets/mvs_shape.py reads (reference mvs_shape.py:28-121): everything of a view
    in <root>/<split>_%03d/ — metadata.json with cam_loc / imh / imw only, the four surface buffers, rgba.png for
    train / val — plus <root>/lights.npz (models/shape.py:63-69).  Coordinates in `xyz_units` (DTU scenes are
    millimetre-sized; the MVS configs set xyz_scale = 1e-3)."""
    rng = np.random.default_rng(seed)
    light_xyz, light_areas = nerfactor_ref.gen_light_xyz(light_h, 2 * light_h)
    os.makedirs(root, exist_ok=True)
    np.savez(join(root, 'lights.npz'), lxyzs=(light_xyz * xyz_units).astype(np.float32), lareas=light_areas.astype(np.float32))
    for split, n in (('train', n_train), ('val', n_val), ('test', n_test)):
        for i in range(n):
            v = rng.normal(size=3)
            v[2] = abs(v[2]) + 0.3
            cam_loc = 4. * v / np.linalg.norm(v)
            _, rgb, _, alpha, xyz, normal, lvis = _view(cam_loc, imh, imw, light_xyz.reshape(-1, 3))
            view = join(root, '%s_%03d' % (split, i))
            os.makedirs(view, exist_ok=True)
            with open(join(view, 'metadata.json'), 'w') as h:
                json.dump({'cam_loc': [float(x) * xyz_units for x in cam_loc], 'imh': imh, 'imw': imw}, h)
            if split != 'test':
                _png(join(view, 'rgba.png'), np.concatenate((rgb, alpha[..., None]), -1))
            _png(join(view, 'alpha.png'), alpha)
            np.save(join(view, 'xyz.npy'), (xyz * xyz_units).astype(np.float32))
            np.save(join(view, 'normal.npy'), (normal * 1.7).astype(np.float32))     # un-normalised on disk: the loader re-normalises
            np.save(join(view, 'lvis.npy'), (lvis * 1.2 - 0.1).astype(np.float32))   # out of [0, 1] on disk: the loader clips
    return root


def write_merl(root, names=('alum-bronze', 'blue_rubber', 'gold-metallic-paint'), n_rows=4096, seed=0):
    """Tiny stand-in for the pre-processed MERL tables (datasets/brdf_merl.py layout): <root>/{train,vali}_<name>.npz
    and one test.npz; reflectance = a smooth positive lobe of the Rusinkiewicz angles, different per material."""
    rng = np.random.default_rng(seed)
    os.makedirs(root, exist_ok=True)

    def coords(n):
        return np.stack((rng.uniform(0, np.pi, n), rng.uniform(0, np.pi / 2, n), rng.uniform(0, np.pi / 2, n)),
                        1).astype(np.float32)
    for i, name in enumerate(names):
        for split, n in (('train', n_rows), ('vali', n_rows // 4)):
            rus = coords(n)
            refl = (0.05 + 0.1 * i + (1. + i) * np.exp(-(4. + 2 * i) * rus[:, 1:2] ** 2)
                    + 0.02 * np.cos(rus[:, 2:3])).astype(np.float32)
            np.savez(join(root, '%s_%s.npz' % (split, name)), name=name, i=np.int32(i), envmap_h=np.int32(16),
                     ims=np.int32(128), spp=np.int32(1), rusink=rus, refl=refl)
    np.savez(join(root, 'test.npz'), envmap_h=np.int32(16), ims=np.int32(128), spp=np.int32(1), rusink=coords(512))
    return list(names)
