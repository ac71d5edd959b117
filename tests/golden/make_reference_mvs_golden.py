"""Generates tests/golden/reference_mvs.npz: the REFERENCE's nerfactor/datasets/mvs_shape.py (unmodified, imported from
/root/reference, TensorFlow calls on tests/golden/tf_shim) reading a synthetic MVS-layout scene
(tests/synth_scene.py:write_mvs_scene) — `_glob`, `_load_data`, `_process_example_postcache` in vali and test mode.

    python tests/golden/make_reference_mvs_golden.py        (build container only: needs /root/reference)

tests/test_cpu_reference_golden.py holds nerfactor_amd/nerfactor/datasets/mvs_shape.py to these batches.
"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('NERFACTOR_REFERENCE', '/root/reference')
sys.path[:0] = [os.path.join(HERE, 'tf_shim'), REF, os.path.join(REF, 'nerfactor'), REPO]

import tensorflow as tf  # noqa: E402  (the shim)

assert 'numpy-shim' in tf.__version__
from nerfactor.util import io as ioutil  # noqa: E402
from nerfactor.datasets.mvs_shape import Dataset as MvsDataset  # noqa: E402

from tests import synth_scene  # noqa: E402
from tests.golden import golden_inputs as gi  # noqa: E402


def main():
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        root = synth_scene.write_mvs_scene(os.path.join(tmp, 'mvs'), **gi.MVS_SCENE_KW)
        for ini in ('shape_mvs.ini', 'nerfactor_mvs.ini'):
            cfg = ioutil.read_config(os.path.join(REF, 'nerfactor', 'config', ini))
            assert cfg.get('DEFAULT', 'dataset') == 'mvs_shape'
        cfg.set('DEFAULT', 'mvs_root', root)
        cfg.set('DEFAULT', 'imh', str(gi.MVS_SCENE_KW['imh']))
        for mode in ('vali', 'test'):
            ds = MvsDataset(cfg, mode)
            id_, *arrays = ds._load_data(ds.files[0])
            batch = ds._process_example_postcache(
                id_, *(tf.convert_to_tensor(np.asarray(a, np.float32)) for a in arrays))
            assert str(np.asarray(batch[0])[0]) == ('val_000' if mode == 'vali' else 'test_000')
            out['mvs_%s_hw' % mode] = np.asarray(batch[1])
            for k, v in zip(('rayo', 'rayd', 'rgb', 'alpha', 'xyz', 'normal', 'lvis'), batch[2:]):
                a = np.asarray(v)
                assert a.dtype == np.float32, (k, a.dtype)
                out['mvs_%s_%s' % (mode, k)] = a
        out['mvs_n_train_views'] = np.int32(len(MvsDataset(cfg, 'train').files))
        out['mvs_xyz_scale'] = np.float32(cfg.getfloat('DEFAULT', 'xyz_scale'))
    path = os.path.join(HERE, 'reference_mvs.npz')
    np.savez_compressed(path, **out)
    print('wrote %s (%.1f KiB)' % (path, os.path.getsize(path) / 1024))
    for k, v in out.items():
        print('  %-24s %-16s %s' % (k, v.shape, v.dtype))


if __name__ == '__main__':
    main()
