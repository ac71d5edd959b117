"""Surface geometry from a trained NeRF — the stage between NeRF and NeRFactor, with the reference's command line and
outputs (nerfactor/geometry_from_nerf.py:30-391):

    [torchrun --nproc-per-node N] python -m nerfactor_amd.nerfactor.geometry_from_nerf \\
        --trained_nerf=<outroot>/<xname> --out_root=<dir> [--imh H] [--scene_bbox x0,x1,y0,y1,z0,z1]
        [--lvis_far 1] [--occu_thres 0] [--light_h 16] [--spp 1] [--debug]

For every view of every split writes <out_root>/<view id>/{alpha.png, xyz.npy, xyz.png, normal.npy, normal.png,
lvis.npy, lvis.png} — what datasets/nerf_shape.py reads.  All marching runs on libnfx:
  * camera rays: 64 + n_samples_coarse (= 128) coarse density samples -> inverse-CDF -> + 64 + n_samples_fine (= 192)
    samples, all 320 evaluated by the fine network;
    density AND its spatial gradient from ONE fused kernel (nfx_nerf_sigma_grad; the reference differentiates through
    the network with GradientTape.batch_jacobian); expected depth / normal from the compositing weights;
  * shadow rays: every (surface point, front-lit light) pair is a ray from lvis_near = 0.1 to lvis_far marched with the
    density-only kernel (nfx_nerf_sigma_fwd), lvis = 1 - sum(weights).
With N ranks the rays of EVERY view are split into N contiguous ranges: each rank marches its rays and
writes its rows of the .npy files; rank 0 receives uint8 preview rows only."""
import argparse
import glob
import os
import re
import sys
from os.path import basename, exists, join

import numpy as np
import torch
from PIL import Image

from .. import dist as nfx_dist
from ..brdf.renderer import gen_light_xyz
from . import datasets, models
from .util import config as configutil


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--trained_nerf', required=True, help="trained NeRF up to (and including) the learning rate folder")
    ap.add_argument('--data_root', default='', help="input data root (defaults to the one NeRF was trained on)")
    ap.add_argument('--out_root', required=True, help="output root")
    ap.add_argument('--imh', type=int, default=None, help="image height (defaults to NeRF training's)")
    ap.add_argument('--scene_bbox', default=None, help="x_min,x_max,y_min,y_max,z_min,z_max")
    ap.add_argument('--lvis_far', type=float, default=1., help="far plane for tracing light visibility")
    ap.add_argument('--occu_thres', type=float, default=0., help="occupancy threshold surface points have to pass")
    ap.add_argument('--light_h', type=int, default=16)
    ap.add_argument('--mlp_chunk', type=int, default=1 << 25, help="density samples per kernel launch")
    ap.add_argument('--lpix_chunk', type=int, default=1, help="accepted for compatibility (pairs are batched by mlp_chunk)")
    ap.add_argument('--spp', type=int, default=1, help="samples per pixel")
    ap.add_argument('--fps', type=int, default=12, help="accepted for compatibility (no video is written)")
    ap.add_argument('--debug', action='store_true')
    return ap.parse_args(argv)


def latest_checkpoint(trained_nerf):
    """Highest-numbered checkpoint prefix: a torch file `ckpt-N` of this framework or a TensorFlow `ckpt-N.index`."""
    found = set()
    for p in glob.glob(join(trained_nerf, 'checkpoints', 'ckpt-*')):
        m = re.fullmatch(r'(ckpt-\d+)(\.index)?', basename(p))
        if m:
            found.add(join(trained_nerf, 'checkpoints', m.group(1)))
    if not found:
        raise FileNotFoundError("no checkpoint under %s" % join(trained_nerf, 'checkpoints'))
    return max(found, key=lambda p: int(p.rsplit('-', 1)[1]))


def _march(model, rayo, rayd, near, far, n_coarse, n_fine, lin_in_disp, bbox, want_normal, rays_per_call):
    """Coarse + importance-sampled density march of `rayo + rayd z`, z in [near, far]; returns (occu, depth, normal)."""
    n = rayo.shape[0]
    occu = torch.empty(n, device=rayo.device)
    depth = torch.empty_like(occu)
    normal = torch.empty((n, 3), device=rayo.device) if want_normal else None
    for lo in range(0, n, rays_per_call):
        o, d = rayo[lo:lo + rays_per_call].contiguous(), rayd[lo:lo + rays_per_call].contiguous()
        z = model.gen_z(near, far, n_coarse, o.shape[0], lin_in_disp=lin_in_disp, perturb=False, device=o.device)
        w = model.accumulate_sigma(model.eval_sigma(o, d, z, use_fine=False, bbox=bbox), z, d)
        z = model.gen_z_fine(z, w, n_fine, perturb=False)
        if want_normal:
            sigma, nrm = model.eval_sigma_normal(o, d, z, bbox=bbox)
        else:
            sigma, nrm = model.eval_sigma(o, d, z, use_fine=True, bbox=bbox), None
        w = model.accumulate_sigma(sigma, z, d)
        occu[lo:lo + rays_per_call] = w.sum(-1)
        depth[lo:lo + rays_per_call] = (w * z).sum(-1)
        if want_normal:
            normal[lo:lo + rays_per_call] = (w[:, :, None] * nrm).sum(1)
    return occu, depth, normal


def _sample_counts(config):
    return (64 + config.getint('DEFAULT', 'n_samples_coarse'), 64 + config.getint('DEFAULT', 'n_samples_fine'),
            config.getboolean('DEFAULT', 'lin_in_disp'))


def compute_depth_and_normal(model, rayo, rayd, config, bbox=None, mlp_chunk=1 << 25):
    """(occu[N], exp_depth[N], exp_normal[N,3]) — geometry_from_nerf.py:249-319."""
    n_coarse, n_fine, lin_in_disp = _sample_counts(config)
    rays = max(1, mlp_chunk // (n_coarse + n_fine))
    return _march(model, rayo, rayd, config.getfloat('DEFAULT', 'near'), config.getfloat('DEFAULT', 'far'), n_coarse,
                  n_fine, lin_in_disp, bbox, True, rays)


def compute_light_visibility(model, surf, normal, config, lvis_far=1., light_h=16, bbox=None, lvis_near=.1,
                             mlp_chunk=1 << 25):
    """lvis[n_surf, n_lights] = 1 - occupancy along surface -> light, 0 for back-lit pairs (:177-246)."""
    n_coarse, n_fine, lin_in_disp = _sample_counts(config)
    lxyz, _ = gen_light_xyz(light_h, 2 * light_h)
    lxyz = torch.as_tensor(lxyz.reshape(-1, 3).astype(np.float32), device=surf.device)
    n, n_lights = surf.shape[0], lxyz.shape[0]
    lvis = torch.zeros((n, n_lights), device=surf.device)
    pts_per_call = max(1, mlp_chunk // ((n_coarse + n_fine) * n_lights))
    rays_per_call = max(1, mlp_chunk // (n_coarse + n_fine))
    for lo in range(0, n, pts_per_call):
        s = surf[lo:lo + pts_per_call]
        surf2l = torch.nn.functional.normalize(lxyz[None] - s[:, None], dim=2, eps=1e-12)   # tf.math.l2_normalize
        front = (surf2l * normal[lo:lo + pts_per_call, None]).sum(-1) > 0
        if not bool(front.any()):
            continue
        o = s[:, None, :].expand(-1, n_lights, -1)[front]
        occu, _, _ = _march(model, o, surf2l[front], lvis_near, lvis_far, n_coarse, n_fine, lin_in_disp, bbox, False,
                            rays_per_call)
        block = lvis[lo:lo + pts_per_call]
        block[front] = 1. - occu
    return lvis


def average_supersamples(t, sps):
    return torch.stack([t[i::sps, j::sps] for i in range(sps) for j in range(sps)]).mean(0)


def _alpha_blend(a, alpha, bg=None):
    alpha = alpha[..., None] if a.ndim == 3 else alpha
    return a * alpha + (0. if bg is None else bg * (1. - alpha))


def _write_png(path, arr):
    arr = (np.clip(arr, 0, 1) * 255 + .5).astype(np.uint8)
    Image.fromarray(arr).save(path)


def process_view(config, model, batch, args, bbox):
    sps = int(np.sqrt(args.spp))
    id_, hw, rayo, rayd, _ = batch
    id_ = id_[0]
    h, w = int(hw[0, 0]), int(hw[0, 1])
    out_dir = join(args.out_root, id_)
    expected = [join(out_dir, f) for f in ('alpha.png', 'lvis.npy', 'lvis.png', 'normal.npy', 'normal.png', 'xyz.npy',
                                           'xyz.png')]
    if all(exists(x) for x in expected):
        print("[geometry_from_nerf] Skipping %s since it's done already" % id_, flush=True)
        return out_dir
    if sps != 1:
        raise NotImplementedError("spp > 1: the reference's light-visibility masking assumes one ray per pixel")
    os.makedirs(out_dir, exist_ok=True)
    # rays within the view are split into contiguous ranges over the ranks: every step below is per
    # ray at spp = 1, each rank writes its own rows of the .npy files, rank 0 receives uint8 preview rows only
    rank, ws = nfx_dist.world()
    lo, hi = nfx_dist.shard_range(h * w, rank, ws)
    rayo, rayd = rayo[lo:hi], rayd[lo:hi]
    rayd = torch.nn.functional.normalize(rayd, dim=1, eps=1e-12)
    # ------ camera -> object
    occu, exp_depth, exp_normal = compute_depth_and_normal(model, rayo, rayd, config, bbox, args.mlp_chunk)
    occu = torch.where(occu < args.occu_thres, torch.zeros_like(occu), occu)
    alpha = occu.clamp(0., 1.)                                   # average_supersamples is the identity at spp = 1
    surf = rayo + rayd * exp_depth[:, None]
    xyz = _alpha_blend(surf, alpha[:, None])
    bg = exp_normal.new_tensor((0., 1., 0.))   # (0, 0, 0) would give (0, 0, 0) tangents
    normal = torch.nn.functional.normalize(_alpha_blend(exp_normal, alpha[:, None], bg), dim=1, eps=1e-12)
    normal = normal.clamp(-1., 1.)
    # ------ object -> light (the reference masks the per-sample buffers with the averaged alpha: spp = 1 layouts)
    hit = alpha > 0.
    n_lights = 2 * args.light_h * args.light_h
    lvis = torch.zeros((hi - lo, n_lights), device=rayo.device)
    if bool(hit.any()):
        lvis_hit = compute_light_visibility(model, surf[hit], exp_normal[hit], config, lvis_far=args.lvis_far,
                                            light_h=args.light_h, bbox=bbox, mlp_chunk=args.mlp_chunk)
        lvis[hit] = lvis_hit.clamp(0., 1.)
    lvis = lvis * alpha[:, None]
    # ------ writers (util/geom.py:27-79): .npy rows by every rank, previews by rank 0
    shapes = {'xyz': (h, w, 3), 'normal': (h, w, 3), 'lvis': (h, w, n_lights)}
    if rank == 0:
        for name, shp in shapes.items():
            np.lib.format.open_memmap(join(out_dir, name + '.npy.part'), mode='w+', dtype=np.float32, shape=shp).flush()
    nfx_dist.barrier()
    for name, t in (('xyz', xyz), ('normal', normal), ('lvis', lvis)):
        mm = np.lib.format.open_memmap(join(out_dir, name + '.npy.part'), mode='r+')
        mm.reshape(h * w, -1)[lo:hi] = t.cpu().numpy()
        mm.flush()
        del mm
    xyz_min, xyz_max = _global_min_max(xyz)
    span = xyz_max - xyz_min
    q8 = lambda t: (t.clamp(0., 1.) * 255 + .5).to(torch.uint8)
    rows = {'alpha': q8(alpha[:, None]), 'xyz': q8((xyz - xyz_min) / (span if span > 0 else 1.)),
            'normal': q8((normal + 1) / 2), 'lvis': q8(lvis.mean(1, keepdim=True))}
    rows = {k: nfx_dist.gather_cat(v) for k, v in sorted(rows.items())}
    nfx_dist.barrier()
    if rank == 0:
        for name in shapes:
            os.replace(join(out_dir, name + '.npy.part'), join(out_dir, name + '.npy'))
        for name, v in rows.items():
            img = v.cpu().numpy().reshape(h, w, -1)
            Image.fromarray(img[:, :, 0] if img.shape[2] == 1 else img).save(join(out_dir, name + '.png'))
    nfx_dist.barrier()
    return out_dir


def _global_min_max(t):
    """(min, max) of a tensor over all ranks, as Python floats."""
    import torch.distributed as dist
    lo_hi = torch.stack((t.min() if t.numel() else t.new_tensor(float('inf')),
                         -t.max() if t.numel() else t.new_tensor(float('inf'))))
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(lo_hi, op=dist.ReduceOp.MIN)
    return float(lo_hi[0]), -float(lo_hi[1])


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("geometry_from_nerf needs an MI355X: libnfx has no CPU path")
    if int(np.sqrt(args.spp)) ** 2 != args.spp:
        raise ValueError("Samples per pixel must be a square number")
    device = nfx_dist.local_device()
    rank, ws = nfx_dist.init_from_env(device=device)
    ckpt = latest_checkpoint(args.trained_nerf)
    config = configutil.read_config(configutil.get_config_ini(ckpt))
    if args.imh is not None:
        config.set('DEFAULT', 'imh', str(args.imh))
    if args.data_root:
        config.set('DEFAULT', 'data_root', args.data_root)
    bbox = None
    if args.scene_bbox:
        bbox = [float(x) for x in args.scene_bbox.split(',')]
        if len(bbox) != 6:
            raise ValueError("scene_bbox: x_min,x_max,y_min,y_max,z_min,z_max")
    Model = models.get_model_class(config.get('DEFAULT', 'model'))
    model = Model(config).to(device)
    configutil.restore_model(model, ckpt)
    model.to(device)
    Dataset = datasets.get_dataset_class(config.get('DEFAULT', 'dataset'))
    done, i = [], 0
    with torch.no_grad():
        for mode in ('train', 'vali', 'test'):
            try:
                dataset = Dataset(config, mode, always_all_rays=True, spp=args.spp, device=device)
            except FileNotFoundError:
                continue
            for batch in dataset.build_pipeline(no_batch=config.getboolean('DEFAULT', 'no_batch'), no_shuffle=True):
                done.append(process_view(config, model, batch, args, bbox))   # every rank: its rays of this view
                i += 1
                if args.debug:
                    break
    nfx_dist.barrier()
    return done


if __name__ == '__main__':
    main(sys.argv[1:])
