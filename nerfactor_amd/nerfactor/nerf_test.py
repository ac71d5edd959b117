"""Novel-view rendering with a trained NeRF, command line and output layout of the reference
(nerfactor/nerf_test.py:33-84):

    [torchrun --nproc-per-node N] python -m nerfactor_amd.nerfactor.nerf_test --ckpt=<outdir>/checkpoints/ckpt-N

Reads <outdir>.ini, renders every test camera into <outdir>/vis_test/ckpt-N/batch%09d/.  With N ranks every view's rays
are split into N contiguous ranges (util/shard.py): each rank renders and quantises its range, rank 0 receives uint8
rows and writes the images — no data-path collective, and a single view already uses every GPU."""
import argparse
import glob
import sys
from os.path import basename, join

import torch

from .. import dist as nfx_dist
from . import datasets, models
from .util import config as configutil, shard


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--ckpt', required=True, help="path to checkpoint")
    ap.add_argument('--debug', action='store_true', help="debug mode switch")
    return ap.parse_args(argv)


def setup(ckpt, debug, device):
    config_ini = configutil.get_config_ini(ckpt)
    config = configutil.read_config(config_ini)
    outroot = join(config_ini[:-4], 'vis_test', basename(ckpt))
    Dataset = datasets.get_dataset_class(config.get('DEFAULT', 'dataset'))
    dataset = Dataset(config, 'test', debug=debug, device=device)
    datapipe = dataset.build_pipeline(no_batch=config.getboolean('DEFAULT', 'no_batch'), no_shuffle=True)
    Model = models.get_model_class(config.get('DEFAULT', 'model'))
    model = Model(config, debug=debug).to(device)
    configutil.restore_model(model, ckpt)
    model.to(device)
    return config, outroot, dataset, datapipe, model


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("nerf_test needs an MI355X: libnfx has no CPU path")
    device = nfx_dist.local_device()
    rank, ws = nfx_dist.init_from_env(device=device)
    _, outroot, dataset, datapipe, model = setup(args.ckpt, args.debug, device)
    for batch_i, batch in enumerate(datapipe):
        shard.render_view(model, batch, join(outroot, 'batch{i:09d}'.format(i=batch_i)), mode='test')
        if args.debug:
            break
    nfx_dist.barrier()
    view_at = None
    if rank == 0:
        view_at = model.compile_batch_vis(sorted(glob.glob(join(outroot, 'batch?????????'))), outroot, mode='test')
        print("[nerf_test] Compilation available for viewing at\n\t%s" % view_at, flush=True)
    return outroot


if __name__ == '__main__':
    main(sys.argv[1:])
