"""CPU tests of the host-side data readers and driver plumbing (datasets/*, trainvali helpers, ragged gather):
no libnfx compute involved."""
import json
import os
from os.path import exists, join

import numpy as np
import pytest
import torch

from nerfactor_amd.nerfactor import trainvali
from nerfactor_amd.nerfactor.config import make_config
from nerfactor_amd.nerfactor.datasets import get_dataset_class
from oracle import nerf_ref
from tests import synth_scene
from tests.mp_util import run_workers


@pytest.fixture(scope='module')
def scene(tmp_path_factory):
    root = str(tmp_path_factory.mktemp('scene'))
    data_root, nerf_root = synth_scene.write_scene(root, imh=16, imw=16)
    return root, data_root, nerf_root


def _cfg(name, scene, **kw):
    root, data_root, nerf_root = scene
    return make_config(name, data_root=data_root, data_nerf_root=nerf_root, outroot=join(root, 'out'), imh=16,
                       n_rays_per_step=32, **kw)


def test_nerf_dataset_rays_and_compositing(scene):
    cfg = _cfg('nerf', scene)
    ds = get_dataset_class('nerf')(cfg, 'vali', device='cpu')
    assert ds.get_n_views() == 1 and ds.bs == 256
    id_, hw, rayo, rayd, rgb = next(iter(ds.build_pipeline(no_batch=True)))
    assert id_ == ['val_000'] * 256 and hw.dtype == torch.int32 and hw[0].tolist() == [16, 16]
    with open(join(scene[1], 'val_000', 'metadata.json')) as h:
        meta = json.load(h)
    c2w = np.array([float(x) for x in meta['cam_transform_mat'].split(',')]).reshape(4, 4)
    ro, rd = nerf_ref.gen_rays(c2w, meta['cam_angle_x'], 16, 16)
    np.testing.assert_allclose(rayd.numpy(), rd.reshape(-1, 3), atol=1e-6)
    np.testing.assert_allclose(rayo.numpy(), ro.reshape(-1, 3), atol=1e-6)
    # white background: pixels outside the sphere are exactly 1
    d = rd.reshape(-1, 3) / np.linalg.norm(rd.reshape(-1, 3), axis=1, keepdims=True)
    b = (ro.reshape(-1, 3) * d).sum(1)
    miss = b * b - ((ro.reshape(-1, 3) ** 2).sum(1) - 1.) <= 0
    assert miss.any() and (rgb.numpy()[miss] == 1.).all()
    assert rgb.min() >= 0 and rgb.max() <= 1


def test_nerf_dataset_train_sampling_and_test_mode(scene):
    cfg = _cfg('nerf', scene)
    tr = get_dataset_class('nerf')(cfg, 'train', device='cpu')
    assert tr.bs == 32 and tr.get_n_views() == 3
    batches = list(tr.build_pipeline(no_batch=True, seed=1))
    assert len(batches) == 3 and all(b[2].shape == (32, 3) and b[4].shape == (32, 3) for b in batches)
    assert sorted(b[0][0] for b in batches) == ['train_000', 'train_001', 'train_002']
    te = get_dataset_class('nerf')(cfg, 'test', device='cpu')   # cameras without images
    assert te.get_n_views() == 2
    b = next(iter(te.build_pipeline(no_batch=True, no_shuffle=True)))
    assert b[0][0] == 'test_000' and float(b[4].abs().max()) == 0.
    with pytest.raises(ValueError):
        get_dataset_class('nerf')(cfg, 'bogus', device='cpu')
    with pytest.raises(ValueError):
        get_dataset_class('nerf')(cfg, 'train', spp=2, device='cpu')


def test_nerf_shape_dataset(scene):
    cfg = _cfg('nerfactor', scene)
    Dataset = get_dataset_class('nerf_shape')
    va = Dataset(cfg, 'vali', device='cpu')
    id_, hw, rayo, rayd, rgb, alpha, xyz, normal, lvis = next(iter(va.build_pipeline(no_batch=True)))
    assert alpha.shape == (256, 1) and xyz.shape == (256, 3) and lvis.shape == (256, 512)
    fg = alpha[:, 0] > 0.9
    assert 20 < int(fg.sum()) < 256
    np.testing.assert_allclose(np.linalg.norm(xyz.numpy()[fg.numpy()], axis=1), 1., atol=1e-5)  # on the sphere
    np.testing.assert_allclose(np.linalg.norm(normal.numpy(), axis=1), 1., atol=1e-5)
    assert set(np.unique(lvis.numpy())) <= {0., 1.}
    tr = Dataset(cfg, 'train', device='cpu')
    for batch in tr.build_pipeline(no_batch=True):
        assert batch[5].shape == (32, 1) and float(batch[5].min()) > 0.9          # foreground rays only
        assert len(set(batch[0])) == 1                                           # all of one view
    te = Dataset(cfg, 'test', device='cpu')
    assert te.get_n_views() == 2
    assert next(iter(te.build_pipeline(no_batch=True, no_shuffle=True)))[4].abs().max() == 0
    # a view whose buffers are missing is skipped, not an error
    os.rename(join(scene[2], 'train_001', 'lvis.npy'), join(scene[2], 'train_001', 'lvis.bak'))
    try:
        assert Dataset(cfg, 'train', device='cpu').get_n_views() == 2
    finally:
        os.rename(join(scene[2], 'train_001', 'lvis.bak'), join(scene[2], 'train_001', 'lvis.npy'))


def test_config_loading_and_override(tmp_path):
    cfg = trainvali.load_config('shape.ini', 'lr=3e-3,imh=64,xname=a{lr}_h{imh}')
    assert cfg.getfloat('DEFAULT', 'lr') == 3e-3 and cfg.get('DEFAULT', 'model') == 'shape'
    assert cfg.get('DEFAULT', 'xname').format(**trainvali.config2dict(cfg)) == 'a3e-3_h64'
    path = str(tmp_path / 'mine.ini')
    with open(path, 'w') as h:
        cfg.write(h)
    assert trainvali.load_config(path).getint('DEFAULT', 'imh') == 64
    with pytest.raises(FileNotFoundError):
        trainvali.load_config('no_such_model.ini')


def test_checkpoint_manager_and_writer(tmp_path):
    class Opt:
        def __init__(self):
            self.sd = {'iterations': 0}

        def state_dict(self):
            return self.sd

        def load_state_dict(self, sd):
            self.sd = sd
    net, opt = torch.nn.Linear(3, 2), Opt()
    mgr = trainvali.CheckpointManager(str(tmp_path / 'checkpoints'), max_to_keep=2)
    assert mgr.latest_checkpoint is None
    for step in (10, 20, 30):
        opt.sd = {'iterations': step * 3}
        path = mgr.save(net, opt, step)
    assert path.endswith('ckpt-3') and [os.path.basename(p) for p in mgr.all()] == ['ckpt-2', 'ckpt-3']
    net2, opt2 = torch.nn.Linear(3, 2), Opt()
    assert mgr.restore(net2, opt2, mgr.latest_checkpoint) == 30 and opt2.sd['iterations'] == 90
    assert torch.equal(net2.weight, net.weight)
    w = trainvali.ScalarWriter(str(tmp_path / 'summary_train'))
    w.scalar('loss_train', 0.25, 10)
    w.scalar('loss_train', 0.125, 20)
    rows = open(join(str(tmp_path / 'summary_train'), 'scalars.csv')).read().strip().splitlines()
    assert rows == ['step,tag,value', '10,loss_train,0.25', '20,loss_train,0.125']


def test_vis_batch_png_layout(tmp_path):
    from nerfactor_amd.nerfactor.models.base import Model
    m = Model.__new__(Model)
    h, w = 4, 5
    to_vis = {'id': ['val_000'] * (h * w), 'hw': torch.tensor([[h, w]] * (h * w), dtype=torch.int32),
              'pred_rgb': torch.rand(h * w, 3), 'pred_normal': torch.randn(h * w, 3), 'gt_alpha': torch.rand(h * w, 1),
              'pred_lvis': torch.rand(h * w, 8), 'pred_rgb_probes': torch.rand(h * w, 2, 3), 'none': None}
    out = str(tmp_path / 'batch000000000')
    Model.vis_batch(m, to_vis, out, mode='vali', dump_raw_to=str(tmp_path / 'raw.npz'))
    for f in ('pred_rgb.png', 'pred_normal.png', 'gt_alpha.png', 'pred_lvis.png', 'pred_rgb_probes/0001.png',
              'metadata.json'):
        assert exists(join(out, f)), f
    from PIL import Image
    assert Image.open(join(out, 'pred_rgb.png')).size == (w, h)
    assert json.load(open(join(out, 'metadata.json')))['id'] == 'val_000'
    assert np.load(str(tmp_path / 'raw.npz'))['pred_lvis'].shape == (h * w, 8)
    assert Model.compile_batch_vis(m, [out], str(tmp_path / 'all'), mode='vali') == str(tmp_path / 'all.txt')


def _gather_worker(rank, world):
    import torch.distributed as dist
    from nerfactor_amd import dist as nfx_dist
    from nerfactor_amd.nerfactor.trainvali import gather_vis, shard_batch
    n = 11
    full = (['v'] * n, torch.arange(n * 3, dtype=torch.float32).reshape(n, 3))
    ids, x = shard_batch(full)
    lo, hi = nfx_dist.shard_range(n, rank, world)
    assert len(ids) == hi - lo and torch.equal(x, full[1][lo:hi])
    out = gather_vis({'id': ids, 'x': x * 2})
    if rank == 0:
        assert out['id'] == full[0] and torch.equal(out['x'], full[1] * 2)
    assert abs(nfx_dist.sum_over_ranks(torch.tensor(float(rank + 1))) - world * (world + 1) / 2) < 1e-6
    nfx_dist.barrier()


def test_shard_and_ragged_gather_two_ranks():
    run_workers(_gather_worker, 2)


def test_brdf_merl_dataset_and_prior_model_on_cpu(tmp_path):
    """datasets/brdf_merl.py batches and the construction of models/brdf.py (its call needs the GPU: tests/test_gpu_train.py)."""
    from nerfactor_amd.nerfactor.models import get_model_class
    root = str(tmp_path / 'merl')
    names = synth_scene.write_merl(root)
    cfg = make_config('brdf', data_root=root, n_rays_per_step=64)
    Dataset = get_dataset_class('brdf_merl')
    tr = Dataset(cfg, 'train', device='cpu')
    assert tr.brdf_names == sorted(names) and tr.get_n_brdfs() == 3 and tr.bs == 64
    batches = list(tr.build_pipeline(no_batch=True, seed=0))
    assert len(batches) == 3 and all(b[5].shape == (64, 3) and b[6].shape == (64, 1) for b in batches)
    assert sorted(int(b[1][0]) for b in batches) == [0, 1, 2] and batches[0][1].dtype == torch.int32
    va = next(iter(Dataset(cfg, 'vali', device='cpu').build_pipeline(no_batch=True)))
    assert va[5].shape == (1024, 3) and va[0][0] == 'alum-bronze'
    te = Dataset(cfg, 'test', device='cpu', n_iden=3, n_between=3)
    ids = [b[0][0] for b in te.build_pipeline(no_batch=True, no_shuffle=True)]
    assert ids[:3] == sorted(names) and len(ids) == 3 + 2 * 3 and ids[3].startswith('000000_1.000000_')
    torch.manual_seed(0)
    model = get_model_class('brdf')(cfg)
    assert model.brdf_names == sorted(names) and model.latent_code.z.shape == (3, 3)
    # the prior runs on the fused width-128 template of libnfx (row f-4): there is no CPU path, it fails loudly
    from nerfactor_amd._capi import NfxError
    with pytest.raises((NfxError, RuntimeError)):
        model(batches[0], mode='train')
    # interpolated identities of the test split parse back into (material, weight) pairs
    tb = [b for b in te.build_pipeline(no_batch=True, no_shuffle=True)][4]
    assert int(tb[1][0]) == -1
    _, w1, rest = tb[0][0].split('_', 2)
    m1, w2, m2 = model._split_interp_id(rest)
    assert m1 in names and m2 in names and abs(float(w1) + float(w2) - 1) < 1e-6


def test_training_batches_say_they_are_foreground_only(scene):
    """datasets/nerf_shape.py draws training rays from alpha > 0.9 (reference nerf_shape.py:102-107) and tags the
    batch's alpha tensor; vali / test batches (whole views) are not tagged; a rank's shard keeps the tag."""
    from nerfactor_amd.nerfactor.datasets.nerf_shape import known_all_foreground, mark_all_foreground
    from nerfactor_amd.nerfactor.util import shard
    cfg = _cfg('shape', scene)
    Dataset = get_dataset_class('nerf_shape')
    tr = next(iter(Dataset(cfg, 'train', device='cpu').build_pipeline(no_batch=True, seed=0)))
    assert known_all_foreground(tr[5]) and bool((tr[5] > 0.9).all()) and tr[5].shape == (32, 1)
    va = next(iter(Dataset(cfg, 'vali', device='cpu').build_pipeline(no_batch=True)))
    assert not known_all_foreground(va[5]) and not bool((va[5] > 0).all())
    assert known_all_foreground(shard.shard_batch(tr)[5])
    assert not known_all_foreground(mark_all_foreground(torch.ones(4, 1))[:2])   # plain slicing drops the tag


def test_prefetched_training_batches_are_the_plain_ones(scene):
    """datasets/base.py read-ahead (the reference's .prefetch, datasets/base.py:110-113): a batch is a pure function of
    (ini seed, mode, pipeline seed, position), so the producer thread running ahead changes nothing — same batches with
    and without it, the same again when an epoch is abandoned half way (its thread ends) and restarted, different rays
    for a different pipeline seed or ini seed."""
    import threading

    def dict_cfg(model, data_root, **kw):
        return make_config(model, data_root=data_root, outroot=join(scene[0], 'out'), n_rays_per_step=64, **kw)
    for name, model in (('nerf_shape', 'shape'), ('nerf', 'nerf'), ('brdf_merl', 'brdf')):
        if name == 'brdf_merl':
            from tests import synth_scene
            root = join(scene[0], 'merl_prefetch')
            synth_scene.write_merl(root)
            cfgs = {p: dict_cfg(model, root, prefetch=p) for p in (0, 2)}
        else:
            cfgs = {p: _cfg(model, scene, prefetch=p) for p in (0, 2)}
        Dataset = get_dataset_class(name)

        def epochs(cfg, seeds=(0, 1)):
            ds = Dataset(cfg, 'train', device='cpu')
            return [[t for t in b if isinstance(t, torch.Tensor)] for s in seeds for b in ds.build_pipeline(no_batch=True, seed=s)]
        plain, ahead = epochs(cfgs[0]), epochs(cfgs[2])
        assert len(plain) == len(ahead) >= 4
        for a, b in zip(plain, ahead):
            assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b))
        n = len(plain) // 2
        assert not all(torch.equal(x, y) for x, y in zip(plain[0], plain[n]))          # another pipeline seed
        ds = Dataset(cfgs[2], 'train', device='cpu')
        first = ds.build_pipeline(no_batch=True, seed=0).take(1)
        assert not [t for t in threading.enumerate() if t.name == 'nfx-prefetch']      # abandoned epoch: thread gone
        again = ds.build_pipeline(no_batch=True, seed=0).take(2)
        for a, b in zip([t for t in first[0] if isinstance(t, torch.Tensor)], [t for t in again[0] if isinstance(t, torch.Tensor)]):
            assert torch.equal(a, b)
        assert all(torch.equal(x, y) for x, y in zip([t for t in again[1] if isinstance(t, torch.Tensor)], plain[1]))
        cfg_other = _cfg(model, scene, seed=7) if name != 'brdf_merl' else dict_cfg(model, root, seed=7)
        other = epochs(cfg_other, seeds=(0,))
        assert not all(torch.equal(x, y) for x, y in zip(other[0], plain[0]))          # another ini seed


def test_prefetch_errors_surface_in_the_consumer(scene, monkeypatch):
    Dataset = get_dataset_class('nerf_shape')
    ds = Dataset(_cfg('shape', scene, prefetch=2), 'train', device='cpu')

    def boom(*a, **k):
        raise RuntimeError("corrupt view")
    monkeypatch.setattr(ds, '_process_example_postcache', boom)
    with pytest.raises(RuntimeError, match="corrupt view"):
        list(ds.build_pipeline(no_batch=True, seed=0))


def test_whole_view_pipelines_read_ahead_and_do_not_hoard_views(scene):
    """Validation / test pipelines: views read ahead by the producer thread (ini prefetch_views) are the same views in
    the same order; only the training dataset keeps every pre-cache result (a test pass visits each 1.4 GB view once)."""
    Dataset = get_dataset_class('nerf_shape')
    views = {}
    for p in (0, 1, 3):
        ds = Dataset(_cfg('shape', scene, prefetch_views=p), 'test', device='cpu')
        views[p] = list(ds.build_pipeline(no_batch=True, no_shuffle=True))
        assert len(ds._cache) == 0 and len(views[p]) == ds.get_n_views() == 2
    for p in (1, 3):
        for a, b in zip(views[0], views[p]):
            assert a[0] == b[0] and all(torch.equal(x, y) for x, y in zip(a[1:], b[1:]))
    va = Dataset(_cfg('shape', scene, prefetch_views=1), 'vali', device='cpu')
    assert len(va.build_pipeline(no_batch=True).take(1)) == 1 and len(va._cache) == 0
    tr = Dataset(_cfg('shape', scene), 'train', device='cpu')
    list(tr.build_pipeline(no_batch=True, seed=0))
    assert len(tr._cache) == tr.get_n_views() and len(tr._candidates) == tr.get_n_views()
    nocache = Dataset(_cfg('shape', scene, cache='false'), 'train', device='cpu')
    list(nocache.build_pipeline(no_batch=True, seed=0))
    assert len(nocache._cache) == 0 and len(nocache._candidates) == 0


def test_write_vis_per_light_images_round_trip(tmp_path):
    """Model.write_vis: [rays, lights, 3] rows (OLAT / probe relighting) become one PNG per light through a single
    transposing pass and the PNG encoder pool; every file holds exactly its light's column."""
    from PIL import Image
    from nerfactor_amd.nerfactor.models.base import Model, _lights_major
    h, w, nl = 5, 7, 9
    rng = np.random.default_rng(3)
    olat = rng.integers(0, 256, size=(h * w, nl, 3), dtype=np.uint8)
    assert np.array_equal(_lights_major(torch.from_numpy(olat)), olat.transpose(1, 0, 2))
    assert np.array_equal(_lights_major(olat[:, ::2]), olat[:, ::2].transpose(1, 0, 2))      # non-contiguous input
    rows = {'id': 'v', 'hw': (h, w), 'pred_rgb_olat': torch.from_numpy(olat),
            'pred_alpha': torch.from_numpy(olat[:, 0, :1].copy()), 'pred_rgb': olat[:, 1].copy(),
            'skipped': torch.zeros(3, 3, dtype=torch.uint8)}
    Model.write_vis(rows, str(tmp_path))
    for i in range(nl):
        got = np.asarray(Image.open(str(tmp_path / 'pred_rgb_olat' / ('%04d.png' % i))))
        assert np.array_equal(got, olat[:, i].reshape(h, w, 3))
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / 'pred_alpha.png'))), olat[:, 0, 0].reshape(h, w))
    assert np.array_equal(np.asarray(Image.open(str(tmp_path / 'pred_rgb.png'))), olat[:, 1].reshape(h, w, 3))
    assert not (tmp_path / 'skipped.png').exists()
