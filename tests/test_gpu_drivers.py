"""End-to-end driver tests on the GPU: the reference's three-stage NeRFactor workflow on a tiny synthetic scene
(shape pre-training -> joint optimisation -> test/relight/edit) through trainvali.py / test.py / nerf_test.py,
checking the output layout of nerfactor/trainvali.py:64-256 and nerfactor/test.py:143-199."""
import csv
import glob
import os
from os.path import exists, join

import numpy as np
import pytest
import torch

from tests import synth_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def scene(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('scene'))
    data_root, nerf_root = synth_scene.write_scene(root, imh=24, imw=24, n_train=3, n_val=1, n_test=3)
    return root, data_root, nerf_root


def _override(scene, **kw):
    root, data_root, nerf_root = scene
    kv = dict(data_root=data_root, data_nerf_root=nerf_root, imh=24, n_rays_per_step=128, vali_batches=1,
              vis_train_batches=1)
    kv.update(kw)
    return ','.join('%s=%s' % item for item in kv.items())


def _scalars(path):
    with open(path) as h:
        return [(int(r['step']), r['tag'], float(r['value'])) for r in csv.DictReader(h)]


@pytest.fixture(scope='module')
def shape_run(nfx_lib, cuda, scene):
    from nerfactor_amd.nerfactor import trainvali
    ov = _override(scene, outroot=join(scene[0], 'out_shape'), epochs=6, ckpt_period=3, vali_period=3,
                   use_nerf_alpha=False)
    outdir = trainvali.main(['--config=shape.ini', '--config_override=' + ov])
    return outdir, ov


def test_shape_trainvali_layout_and_descent(shape_run):
    outdir, _ = shape_run
    assert outdir.endswith('out_shape/lr1e-2') and exists(outdir + '.ini')
    assert sorted(os.listdir(join(outdir, 'checkpoints'))) == ['ckpt-1', 'ckpt-2']
    tr = _scalars(join(outdir, 'summary_train', 'scalars.csv'))
    assert [s for s, t, _ in tr if t == 'loss_train'] == [3, 6]
    losses = [v for _, t, v in tr if t == 'loss_train']
    assert np.isfinite(losses).all()   # (descent on a FIXED batch is pinned in test_gpu_train.py)
    va = _scalars(join(outdir, 'summary_vali', 'scalars.csv'))
    assert [s for s, t, _ in va] == [3, 6] and all(np.isfinite(v) for _, _, v in va)
    bdir = join(outdir, 'vis_vali', 'epoch000000006', 'batch000000000')
    for f in ('pred_normal.png', 'pred_lvis.png', 'gt_normal.png', 'metadata.json'):
        assert exists(join(bdir, f)), f
    assert exists(join(outdir, 'vis_train', 'epoch000000003', 'batch000000000_raw.npz'))


def test_shape_trainvali_resumes(shape_run, scene):
    from nerfactor_amd.nerfactor import trainvali
    outdir, ov = shape_run
    state = torch.load(join(outdir, 'checkpoints', 'ckpt-2'), map_location='cpu')
    assert state['step'] == 6 and int(state['optimizer']['iterations']) == 18      # 3 views x 6 epochs
    trainvali.main(['--config=shape.ini', '--config_override=' + ov.replace('epochs=6', 'epochs=9')])
    assert sorted(os.listdir(join(outdir, 'checkpoints'))) == ['ckpt-1', 'ckpt-2', 'ckpt-3']
    state = torch.load(join(outdir, 'checkpoints', 'ckpt-3'), map_location='cpu')
    assert state['step'] == 9 and int(state['optimizer']['iterations']) == 27
    steps = [s for s, t, _ in _scalars(join(outdir, 'summary_train', 'scalars.csv')) if t == 'loss_train']
    assert steps == [3, 6, 9]


@pytest.fixture(scope='module')
def brdf_run(nfx_lib, cuda, scene):
    """The BRDF prior trained by the same driver on a tiny MERL-like table set (stage 0 of the workflow)."""
    from nerfactor_amd.nerfactor import trainvali
    merl = join(scene[0], 'merl')
    synth_scene.write_merl(merl)
    ov = 'data_root=%s,outroot=%s,epochs=6,ckpt_period=3,vali_period=3,vali_batches=2,n_rays_per_step=512' % (
        merl, join(scene[0], 'out_brdf'))
    outdir = trainvali.main(['--config=brdf.ini', '--config_override=' + ov])
    return outdir


def test_brdf_prior_trainvali(brdf_run):
    losses = [v for _, t, v in _scalars(join(brdf_run, 'summary_train', 'scalars.csv')) if t == 'loss_train']
    assert len(losses) == 2 and np.isfinite(losses).all()   # (random rows of random materials: no monotonicity in 6 epochs)
    state = torch.load(join(brdf_run, 'checkpoints', 'ckpt-2'), map_location='cpu')
    assert state['net']['latent_code._z'].shape == (3, 3)
    assert exists(join(brdf_run, 'vis_vali', 'epoch000000006', 'batch000000001_raw.npz'))


@pytest.mark.parametrize('model', ['nerfactor_microfacet', 'nerfactor'])
def test_joint_optimisation_then_test_driver(shape_run, brdf_run, scene, model):
    from nerfactor_amd.nerfactor import test as test_driver, trainvali
    shape_ckpt = join(shape_run[0], 'checkpoints', 'ckpt-2')
    ov = _override(scene, outroot=join(scene[0], 'out_' + model), epochs=4, ckpt_period=2, vali_period=2,
                   shape_model_ckpt=shape_ckpt, brdf_model_ckpt=join(brdf_run, 'checkpoints', 'ckpt-2'),
                   test_envmap_dir='', shape_mode='finetune')
    outdir = trainvali.main(['--config=%s.ini' % model, '--config_override=' + ov])
    ckpt = join(outdir, 'checkpoints', 'ckpt-2')
    assert exists(ckpt)
    losses = [v for _, t, v in _scalars(join(outdir, 'summary_train', 'scalars.csv')) if t == 'loss_train']
    assert len(losses) == 2 and np.isfinite(losses).all()
    vdir = join(outdir, 'vis_vali', 'epoch000000004', 'batch000000000')
    for f in ('pred_rgb.png', 'pred_albedo.png', 'pred_normal.png', 'gt_rgb.png'):
        assert exists(join(vdir, f)), f

    out = test_driver.main(['--ckpt=' + ckpt, '--color_correct_albedo'])
    assert out == join(outdir, 'vis_test', 'ckpt-2')
    bdirs = sorted(glob.glob(join(out, 'batch?????????')))
    assert len(bdirs) == 3
    assert exists(join(bdirs[0], 'pred_rgb.png')) and not exists(join(bdirs[0], 'pred_rgb_olat'))
    assert len(os.listdir(join(bdirs[2], 'pred_rgb_olat'))) == 512      # OLAT only for the final view
    assert exists(out + '.txt')
    # --debug: only view test_002 (the reference's debug glob), one batch
    out2 = test_driver.main(['--ckpt=' + ckpt, '--tgt_albedo=rainbow', '--sv_axis_i=2', '--sv_axis_min=-1',
                             '--sv_axis_max=1', '--debug'])
    assert out2 == out + '_rainbow' and exists(join(out2, 'batch000000000', 'pred_albedo.png'))
    from PIL import Image
    alb = np.asarray(Image.open(join(out2, 'batch000000000', 'pred_albedo.png')))
    assert len(np.unique(alb.reshape(-1, 3), axis=0)) <= 8 + 1           # 7 bands + background
    if model == 'nerfactor':   # material editing with a latent code of the trained prior
        out3 = test_driver.main(['--ckpt=' + ckpt, '--tgt_brdf=blue_rubber', '--debug'])
        assert out3 == out + '_blue_rubber' and exists(join(out3, 'batch000000000', 'pred_rgb.png'))


def test_nerf_trainvali_then_nerf_test(nfx_lib, cuda, scene):
    """Stage 1 of the reference workflow: trainvali --config=nerf.ini, then nerf_test on its checkpoint."""
    from nerfactor_amd.nerfactor import nerf_test, trainvali
    ov = _override(scene, outroot=join(scene[0], 'out_nerf'), epochs=4, ckpt_period=2, vali_period=4,
                   n_samples_coarse=16, n_samples_fine=32, lr='5e-4')
    outdir = trainvali.main(['--config=nerf.ini', '--config_override=' + ov])
    losses = [v for _, t, v in _scalars(join(outdir, 'summary_train', 'scalars.csv')) if t == 'loss_train']
    assert len(losses) == 2 and np.isfinite(losses).all()
    assert exists(join(outdir, 'vis_vali', 'epoch000000004', 'batch000000000', 'fine_rgb.png'))
    out = nerf_test.main(['--ckpt=' + join(outdir, 'checkpoints', 'ckpt-2')])
    assert len(glob.glob(join(out, 'batch?????????', 'fine_rgb.png'))) == 3
    # ---- stage 2: geometry_from_nerf writes what datasets/nerf_shape.py reads
    from nerfactor_amd.nerfactor import geometry_from_nerf
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.datasets import get_dataset_class
    surf_root = join(scene[0], 'surf')
    done = geometry_from_nerf.main(['--trained_nerf=' + outdir, '--out_root=' + surf_root, '--lvis_far=1',
                                    '--scene_bbox=-1.5,1.5,-1.5,1.5,-1.5,1.5'])
    assert len(done) == 3 + 1 + 3
    vdir = join(surf_root, 'train_000')
    xyz, normal, lvis = (np.load(join(vdir, f + '.npy')) for f in ('xyz', 'normal', 'lvis'))
    assert xyz.shape == (24, 24, 3) and normal.shape == (24, 24, 3) and lvis.shape == (24, 24, 512)
    assert np.isfinite(xyz).all() and np.isfinite(lvis).all() and lvis.min() >= 0 and lvis.max() <= 1
    np.testing.assert_allclose(np.linalg.norm(normal, axis=2), 1., atol=1e-4)
    for f in ('alpha.png', 'xyz.png', 'normal.png', 'lvis.png'):
        assert exists(join(vdir, f)), f
    assert geometry_from_nerf.main(['--trained_nerf=' + outdir, '--out_root=' + surf_root]) == done   # resumable: skips
    cfg = make_config('shape', data_root=scene[1], data_nerf_root=surf_root, imh=24, n_rays_per_step=16)
    batch = next(iter(get_dataset_class('nerf_shape')(cfg, 'vali', device='cpu').build_pipeline(no_batch=True)))
    assert batch[8].shape == (576, 512)


def test_nerf_test_driver(nfx_lib, cuda, scene, tmp_path):
    """A (random-weight) NeRF checkpoint in the trainvali layout renders through nerf_test.py."""
    from nerfactor_amd.nerfactor import nerf_test
    from nerfactor_amd.nerfactor.config import make_config
    from nerfactor_amd.nerfactor.models import get_model_class
    outdir = str(tmp_path / 'out_nerf' / 'lr1e-4')
    cfg = make_config('nerf', data_root=scene[1], imh=24, outroot=str(tmp_path / 'out_nerf'), n_samples_coarse=16,
                      n_samples_fine=32)
    os.makedirs(join(outdir, 'checkpoints'))
    with open(outdir + '.ini', 'w') as h:
        cfg.write(h)
    torch.manual_seed(0)
    model = get_model_class('nerf')(cfg)
    model.register_trainable()
    torch.save({'net': model.state_dict(), 'step': 0}, join(outdir, 'checkpoints', 'ckpt-1'))
    out = nerf_test.main(['--ckpt=' + join(outdir, 'checkpoints', 'ckpt-1')])
    bdirs = sorted(glob.glob(join(out, 'batch?????????')))
    assert len(bdirs) == 3 and exists(join(bdirs[0], 'fine_rgb.png')) and exists(join(bdirs[1], 'coarse_rgb.png'))


def test_two_ranks_reproduce_the_one_process_training_and_render(nfx_lib, cuda, scene):
    """The multi-process paths end to end on this one GPU (NFX_REHEARSAL=1: both ranks on GPU 0, collectives over gloo):
    trainvali — parameters broadcast from rank 0, every batch sharded over the ranks, one [gradients | loss] all-reduce
    per step — and test.py — each view's rays split over the ranks, uint8 rows gathered on rank 0 — against the same
    commands run by one process (jitter off, so the two differ only by fp32 summation order)."""
    import socket
    import subprocess
    import sys
    from PIL import Image
    root = scene[0]

    def run(name, launcher, env):
        ov = _override(scene, outroot=join(root, 'reh_' + name), epochs=3, ckpt_period=3, vali_period=3,
                       use_nerf_alpha=False, shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='',
                       xyz_jitter_std=0, seed=3)
        e = dict(os.environ, **env)
        for mod, arg in (('trainvali', ['--config=nerfactor_microfacet.ini', '--config_override=' + ov]),
                         ('test', ['--ckpt=' + join(root, 'reh_' + name, 'lr5e-3', 'checkpoints', 'ckpt-1')])):
            res = subprocess.run(launcher + ['-m', 'nerfactor_amd.nerfactor.' + mod] + arg, env=e, cwd=os.getcwd(),
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
            assert res.returncode == 0, res.stdout[-3000:]
        return join(root, 'reh_' + name, 'lr5e-3')

    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    one = run('one', [sys.executable], {})
    two = run('two', [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2',
                      '--master-addr=127.0.0.1', '--master-port=%d' % port], {'NFX_REHEARSAL': '1'})
    a = torch.load(join(one, 'checkpoints', 'ckpt-1'), map_location='cpu')['net']
    b = torch.load(join(two, 'checkpoints', 'ckpt-1'), map_location='cpu')['net']
    diff = torch.cat([(a[k].float() - b[k].float()).abs().reshape(-1) for k in a if a[k].dtype.is_floating_point])
    assert float(diff.mean()) < 2e-4 and float(diff.quantile(0.99)) < 5e-3, (float(diff.mean()), float(diff.max()))
    l1 = [v for _, t, v in _scalars(join(one, 'summary_train', 'scalars.csv')) if t == 'loss_train']
    l2 = [v for _, t, v in _scalars(join(two, 'summary_train', 'scalars.csv')) if t == 'loss_train']
    np.testing.assert_allclose(l1, l2, rtol=2e-3)
    ims = sorted(glob.glob(join(one, 'vis_test', 'ckpt-1', 'batch*', 'pred_rgb.png')))
    assert len(ims) == 3
    for f in ims:
        x = np.asarray(Image.open(f)).astype(int)
        y = np.asarray(Image.open(f.replace('reh_one', 'reh_two'))).astype(int)
        assert x.shape == y.shape and np.abs(x - y).max() <= 3, f
