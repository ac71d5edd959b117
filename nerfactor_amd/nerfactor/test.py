"""NeRFactor test driver: view synthesis, relighting (probes + OLAT) and material editing, with the reference's
command line and output layout (nerfactor/test.py:29-200):

    [torchrun --nproc-per-node N] python -m nerfactor_amd.nerfactor.test --ckpt=<outdir>/checkpoints/ckpt-N \\
        [--color_correct_albedo] [--tgt_albedo gold] [--tgt_brdf <merl name>] [--sv_axis_i 0 ...]

With N ranks every view's rays are split into N contiguous ranges (util/shard.py): 4 test views x 8 probes keep 8 GPUs
busy; rank 0 receives uint8 rows only.  OLAT relighting only for the final view, as upstream."""
import argparse
import glob
import json
import sys
from os.path import basename, join

import numpy as np
import torch

from .. import dist as nfx_dist
from .datasets.nerf import load_rgba, resize
from .nerf_test import setup
from .util import config as configutil, shard

RAINBOW = ((0.58, 0, 0.83), (0.29, 0, 0.51), (0, 0, 1), (0, 1, 0), (1, 1, 0), (1, 0.5, 0), (1, 0, 0))
SOLID = {'aluminium': (0.913, 0.921, 0.925), 'gold': (1, 0.843, 0), 'green': (0, 1, 0)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--ckpt', required=True, help="path to checkpoint")
    ap.add_argument('--color_correct_albedo', action='store_true')
    ap.add_argument('--sv_axis_i', type=int, default=0, help="along which axis we do spatially-varying edits")
    ap.add_argument('--sv_axis_min', type=float, default=-1.5)
    ap.add_argument('--sv_axis_max', type=float, default=1.5)
    ap.add_argument('--tgt_albedo', default=None, help="albedo edit name")
    ap.add_argument('--tgt_brdf', default=None, help="BRDF edit name")
    ap.add_argument('--debug', action='store_true')
    return ap.parse_args(argv)


def compute_rgb_scales(ckpt, alpha_thres=0.9):
    """Per-channel least-squares scale matching the predicted albedo of the first validation view of the last
    validated epoch to the ground truth, in linear space (test.py:47-88)."""
    config_ini = configutil.get_config_ini(ckpt)
    config = configutil.read_config(config_ini)
    epoch_dirs = sorted(glob.glob(join(config_ini[:-4], 'vis_vali', 'epoch?????????')))
    batch_dir = sorted(glob.glob(join(epoch_dirs[-1], 'batch?????????')))[0]
    with open(join(batch_dir, 'metadata.json')) as h:
        view = json.load(h)['id']
    pred = load_rgba(join(batch_dir, 'pred_albedo.png'))[:, :, :3] ** 2.2     # undo the display gamma
    gt = load_rgba(join(config.get('DEFAULT', 'data_root'), view, 'albedo.png'))
    gt = resize(gt, pred.shape[0])
    fg = gt[:, :, 3] > alpha_thres
    scales = [float(pred[:, :, c][fg].dot(gt[:, :, c][fg]) / pred[:, :, c][fg].dot(pred[:, :, c][fg]))
              for c in range(3)]
    return torch.tensor(scales, dtype=torch.float32)


def get_albedo_override(name, xyz, axis_i, axis_min, axis_max):
    """Solid colours broadcast over points; 'rainbow' bands along one axis (test.py:91-131)."""
    if name in SOLID:
        return torch.tensor(SOLID[name], dtype=torch.float32, device=xyz.device)
    if name == 'rainbow':
        band = torch.floor((xyz[:, axis_i] - axis_min) / ((axis_max - axis_min) / len(RAINBOW))).long()
        inside = (band >= 0) & (band < len(RAINBOW))
        colors = torch.tensor(RAINBOW, dtype=torch.float32, device=xyz.device)
        out = torch.zeros_like(xyz)
        out[inside] = colors[band[inside]]
        return out
    raise NotImplementedError("Target albedo: %s" % name)


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("test needs an MI355X: libnfx has no CPU path")
    device = nfx_dist.local_device()
    rank, ws = nfx_dist.init_from_env(device=device)
    _, outroot, dataset, datapipe, model = setup(args.ckpt, args.debug, device)
    for suffix in (args.tgt_albedo, args.tgt_brdf):
        if suffix:
            outroot = outroot.rstrip('/') + '_%s' % suffix
    n_views = dataset.get_n_views()

    albedo_scales = None
    if not args.tgt_albedo and args.color_correct_albedo:
        albedo_scales = compute_rgb_scales(args.ckpt)
    brdf_z_override = None
    if args.tgt_brdf:
        brdf = model.brdf_model
        brdf_z_override = brdf.latent_code.z[brdf.brdf_names.index(args.tgt_brdf), :].detach()

    for batch_i, batch in enumerate(datapipe):
        relight_olat = batch_i == n_views - 1
        albedo_override = None
        if args.tgt_albedo:   # evaluated on the whole view, then cut to this rank's rays like the batch itself
            albedo_override = shard.shard_rows(get_albedo_override(
                args.tgt_albedo, batch[6], args.sv_axis_i, args.sv_axis_min, args.sv_axis_max), batch[6].shape[0])
        shard.render_view(
            model, batch, join(outroot, 'batch{i:09d}'.format(i=batch_i)), mode='test', relight_olat=relight_olat,
            relight_probes=True, albedo_scales=albedo_scales, albedo_override=albedo_override,
            brdf_z_override=brdf_z_override)
        if args.debug:
            break
    nfx_dist.barrier()
    if rank == 0:
        view_at = model.compile_batch_vis(sorted(glob.glob(join(outroot, 'batch?????????'))), outroot, mode='test')
        print("[test] Compilation available for viewing at\n\t%s" % view_at, flush=True)
    return outroot


if __name__ == '__main__':
    main(sys.argv[1:])
