"""Training + validation driver with the reference's command line, ini handling and output layout
(nerfactor/trainvali.py:44-256), hosted on torch + libnfx:

    [torchrun --nproc-per-node N] python -m nerfactor_amd.nerfactor.trainvali \\
        --config=shape.ini --config_override='k1=v1,k2=v2' [--debug]

`--config` is a built-in name (nerf.ini, shape.ini, nerfactor.ini, nerfactor_microfacet.ini, brdf.ini — the key sets
of nerfactor/config/*.ini) or the path of any ini file of the reference.  Writes <outroot>/<xname>.ini,
<outroot>/<xname>/{checkpoints/ckpt-N, summary_{train,vali}/scalars.csv, vis_{train,vali}/epoch%09d/batch%09d/}.
One process per GPU: every rank reads the same view, keeps its contiguous ray shard (nfx_dist.shard), gradients
and the loss are summed over ranks in one RCCL all-reduce inside the optimizer (optim.AMSGrad.step).
Differences from the reference, by design: checkpoints are torch.save({'net','optimizer','step'}) files named
ckpt-N (same `<outdir>/checkpoints/ckpt-N` path contract for *_model_ckpt keys); summaries are CSV, not TF events.
"""
import argparse
import csv
import glob
import os
import re
import shutil
import sys
import time
from collections import deque
from os.path import dirname, exists, join

import torch

from .. import dist as nfx_dist
from .. import optim
from . import config as builtin_config
from . import datasets, models
from .util import config as configutil


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="A general training and validation pipeline.")
    ap.add_argument('--config', default='nerf.ini', help="base .ini name or a full path")
    ap.add_argument('--config_override', default='', help="e.g., 'key1=value1,key2=value2'")
    ap.add_argument('--debug', action='store_true', help="debug mode switch")
    ap.add_argument('--device', default='gpu', choices=['gpu'], help="libnfx is gfx950-only: no CPU path")
    return ap.parse_args(argv)


def load_config(config_arg, override=''):
    if exists(config_arg):
        config = configutil.read_config(config_arg)
    else:
        name = os.path.basename(config_arg)
        name = name[:-4] if name.endswith('.ini') else name
        if name not in builtin_config.CONFIGS:
            raise FileNotFoundError(config_arg)
        config = builtin_config.make_config(name)
    if override:
        for kv in override.split(','):
            k, v = kv.split('=', 1)
            config.set('DEFAULT', k, v)
    return config


def config2dict(config):
    return {k: v for k, v in config['DEFAULT'].items()}


def prepare_outdir(outdir, overwrite=False, rank=0):
    """util/io.py:24-33 — refuse a non-empty directory unless overwrite (or resuming from its checkpoints)."""
    if rank == 0:
        if exists(outdir) and overwrite:
            shutil.rmtree(outdir)
        os.makedirs(outdir, exist_ok=True)
    nfx_dist.barrier()


class ScalarWriter:
    """summary_<split>/scalars.csv: step,tag,value."""
    def __init__(self, outdir, enabled=True):
        self.path = join(outdir, 'scalars.csv')
        self.enabled = enabled
        if enabled:
            os.makedirs(outdir, exist_ok=True)
            if not exists(self.path):
                with open(self.path, 'w', newline='') as h:
                    csv.writer(h).writerow(['step', 'tag', 'value'])

    def scalar(self, tag, value, step):
        if self.enabled:
            with open(self.path, 'a', newline='') as h:
                csv.writer(h).writerow([int(step), tag, repr(value)])


class CheckpointManager:
    """<ckptdir>/ckpt-N with max_to_keep, latest = highest N (tf.train.CheckpointManager's contract)."""
    def __init__(self, ckptdir, max_to_keep=None):
        self.dir, self.max_to_keep = ckptdir, max_to_keep

    def all(self):
        paths = [p for p in glob.glob(join(self.dir, 'ckpt-*')) if re.fullmatch(r'ckpt-\d+', os.path.basename(p))]
        return sorted(paths, key=lambda p: int(p.rsplit('-', 1)[1]))

    @property
    def latest_checkpoint(self):
        paths = self.all()
        return paths[-1] if paths else None

    def save(self, model, optimizer, step):
        os.makedirs(self.dir, exist_ok=True)
        n = len(self.all()) + 1
        latest = self.latest_checkpoint
        if latest is not None:
            n = int(latest.rsplit('-', 1)[1]) + 1
        path = join(self.dir, 'ckpt-%d' % n)
        tmp = path + '.tmp'
        torch.save({'net': model.state_dict(), 'optimizer': optimizer.state_dict(), 'step': int(step)}, tmp)
        os.replace(tmp, path)
        if self.max_to_keep:
            for old in self.all()[:-self.max_to_keep]:
                os.remove(old)
        return path

    def restore(self, model, optimizer, path):
        state = torch.load(path, map_location='cpu')
        model.load_state_dict(state['net'])
        optimizer.load_state_dict(state['optimizer'])
        return int(state['step'])


def shard_batch(batch):
    """Contiguous ray shard of this rank for every per-ray field of a flat batch tuple (util/shard.py: keeps the
    dataset's foreground-only tag, so a multi-rank training step skips the torch.nonzero compaction as well)."""
    from .util import shard as shardutil
    return shardutil.shard_batch(batch)


def vali_step(model, batch, global_bs):
    """distributed_vali_step (trainvali.py:301-317): summed per-example loss / global batch size over ranks."""
    with torch.no_grad():
        pred, gt, loss_kwargs, to_vis = model(shard_batch(batch), mode='vali')
        loss_kwargs['keep_batch'] = True
        weighted = model.compute_loss(pred, gt, **loss_kwargs).sum() / global_bs
    return nfx_dist.sum_over_ranks(weighted), gather_vis(to_vis)


def gather_vis(to_vis):
    """Concatenates per-rank shards of the visualisation buffers on rank 0 (strategy.experimental_local_results +
    concat in the reference, trainvali.py:320-330)."""
    rank, ws = nfx_dist.world()
    if ws == 1:
        return to_vis
    out = {}
    for k in sorted(to_vis):
        v = to_vis[k]
        if isinstance(v, torch.Tensor):
            out[k] = nfx_dist.gather_cat(v)
        else:
            parts = nfx_dist.gather_objects(v)
            out[k] = [x for p in parts for x in p] if rank == 0 else v
    return out


def maintain_epoch_queue(queue, new_epoch_dir):
    if queue.maxlen is not None and len(queue) == queue.maxlen:
        old = queue.popleft()
        if exists(old):
            shutil.rmtree(old)
    queue.append(new_epoch_dir)


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise RuntimeError("trainvali needs an MI355X: libnfx has no CPU path")
    device = nfx_dist.local_device()
    rank, ws = nfx_dist.init_from_env(device=device)
    is_main = rank == 0
    log = (lambda *a: print('[trainvali]', *a, flush=True)) if is_main else (lambda *a: None)

    config = load_config(args.config, args.config_override)
    xname = config.get('DEFAULT', 'xname').format(**config2dict(config))
    outdir = join(config.get('DEFAULT', 'outroot'), xname)
    prepare_outdir(outdir, overwrite=config.getboolean('DEFAULT', 'overwrite'), rank=rank)
    log("For results, see:\n\t%s" % outdir)
    if is_main:  # the effective configuration, where get_config_ini() expects it
        with open(outdir.rstrip('/') + '.ini', 'w') as h:
            config.write(h)

    Dataset = datasets.get_dataset_class(config.get('DEFAULT', 'dataset'))
    dataset_train = Dataset(config, 'train', debug=args.debug, device=device)
    global_bs_train = dataset_train.bs
    no_batch = config.getboolean('DEFAULT', 'no_batch')
    seed = config.getint('DEFAULT', 'seed', fallback=0)
    try:
        dataset_vali = Dataset(config, 'vali', debug=args.debug, device=device)
        global_bs_vali = dataset_vali.bs
        vali_batches = dataset_vali.build_pipeline(no_batch=no_batch).take(config.getint('DEFAULT', 'vali_batches'))
    except FileNotFoundError:
        vali_batches = None

    Model = models.get_model_class(config.get('DEFAULT', 'model'))
    torch.manual_seed(seed)   # the same initial draw on every rank ...
    model = Model(config, debug=args.debug).to(device)
    model.register_trainable()
    optimizer = optim.make_optimizer(model, config)

    keep = config.getint('DEFAULT', 'keep_recent_epochs')
    keep = keep if keep > 0 else None
    manager = CheckpointManager(join(outdir, 'checkpoints'), max_to_keep=keep)
    step = 0
    if manager.latest_checkpoint:
        step = manager.restore(model, optimizer, manager.latest_checkpoint)
        log("Resumed from step:\n\t%s" % manager.latest_checkpoint)
    else:
        log("Started from scratch")
    # ... and rank 0's values mirrored explicitly (initial weights, light, latent codes, restored checkpoint,
    # optimizer moments): every replica steps from the same point, as under MirroredStrategy (trainvali.py:259-262)
    nfx_dist.broadcast_model(model, optimizer)
    torch.manual_seed(seed * 9973 + 1 + rank)   # per-rank streams for the jitter / perturbation noise

    writer_train = ScalarWriter(join(outdir, 'summary_train'), is_main)
    writer_vali = ScalarWriter(join(outdir, 'summary_vali'), is_main)
    vis_epoch = {m: join(outdir, 'vis_' + m, 'epoch{e:09d}') for m in ('train', 'vali')}
    queues = {m: deque([], keep) for m in ('train', 'vali')}

    epochs = config.getint('DEFAULT', 'epochs')
    vis_train_batches = config.getint('DEFAULT', 'vis_train_batches')
    ckpt_period = config.getint('DEFAULT', 'ckpt_period')
    vali_period = config.getint('DEFAULT', 'vali_period')

    def visualise(mode, batch_vis, step):
        vis_dirs = []
        for b, to_vis in enumerate(batch_vis):
            epoch_dir = vis_epoch[mode].format(e=step)
            vis_dir = join(epoch_dir, 'batch{b:09d}'.format(b=b))
            model.vis_batch(to_vis, vis_dir, mode=mode, dump_raw_to=join(epoch_dir, 'batch{b:09d}_raw.npz'.format(b=b)))
            vis_dirs.append(vis_dir)
        model.compile_batch_vis(vis_dirs, join(vis_epoch[mode].format(e=step), 'all'), mode=mode)
        maintain_epoch_queue(queues[mode], vis_epoch[mode].format(e=step))

    # extra ini key (not in the reference): hip_graph = true captures the training step in a hipGraph per batch shape
    # (optim.GraphedTrainStep); single-process runs with foreground-tagged batches only, eager otherwise
    if config.getboolean('DEFAULT', 'hip_graph', fallback=False):
        graphed = optim.GraphedTrainStep(model, optimizer, global_bs_train)
        run_step = lambda b: graphed(b)
    else:
        run_step = lambda b: optim.train_step(model, b, optimizer, global_bs_train)

    while step < epochs:
        # ------ one epoch = every training view once, n_rays_per_step rays each ------
        losses, batch_vis, batch_time = [], [], []
        pipe = dataset_train.build_pipeline(no_batch=no_batch, seed=seed + step)  # same order on every rank
        for batch_i, batch in enumerate(pipe):
            t0 = time.time()
            loss, to_vis = run_step(shard_batch(batch))
            losses.append(loss.clone())   # device scalars: no host sync inside the epoch (a graphed step reuses its output)
            batch_time.append(time.time() - t0)
            if batch_i < vis_train_batches:
                batch_vis.append(gather_vis(to_vis))
            if args.debug:
                break
        if not batch_time:
            raise RuntimeError("Dataset is empty")
        model.flush_numerics(block=True)   # every outstanding check_numerics verdict of the epoch
        step += 1

        if step % ckpt_period == 0:
            torch.cuda.synchronize()
            if is_main:
                path = manager.save(model, optimizer, step)
                log("Checkpointed step %d:\n\t%s" % (step, path))
                writer_train.scalar('loss_train', float(torch.stack([l.reshape(()) for l in losses]).mean()), step)
                writer_train.scalar('batch_time_train', sum(batch_time) / len(batch_time), step)
                visualise('train', batch_vis, step)
            nfx_dist.barrier()

        if vali_batches is not None and vali_period > 0 and step % vali_period == 0:
            losses, batch_vis = [], []
            for batch in vali_batches:
                loss, to_vis = vali_step(model, batch, global_bs_vali)
                losses.append(float(loss))
                batch_vis.append(to_vis)
            if is_main:
                writer_vali.scalar('loss_vali', sum(losses) / len(losses), step)
                visualise('vali', batch_vis, step)
            nfx_dist.barrier()
    nfx_dist.barrier()
    return outdir


if __name__ == '__main__':
    main(sys.argv[1:])
