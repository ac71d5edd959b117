"""Gradient bucket of one training step of every model, saved (first call) or compared bit for bit (later calls):
the check for layout / scheduling changes of the backward kernels that must not change a single bit.
    NFX_LIB_PATH=old.so python scripts/grad_identity.py save;  python scripts/grad_identity.py check"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import _capi, optim  # noqa: E402
from nerfactor_amd.nerfactor.config import make_config  # noqa: E402
from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground  # noqa: E402
from nerfactor_amd.nerfactor.models import get_model_class  # noqa: E402

dev = torch.device('cuda:0')
mode = sys.argv[1]
path = 'gpurun_out/grad_identity.pt'
out = {}
for name, n in (('nerfactor_microfacet', 1024), ('nerfactor', 300), ('shape', 37), ('nerf', 200), ('nerf', 1024)):
    for lds in ('0', '1'):
        _capi.set_option('wgrad_lds', int(lds))
        _capi.set_option('wgrad_fused', 0)   # this script compares the stored-activation kernels
        torch.manual_seed(3)
        rng = np.random.default_rng(5)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        extra = dict(shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='') if 'nerfactor' in name else {}
        if name == 'brdf':
            from tests import synth_scene  # noqa: F401
            import tempfile
            with tempfile.TemporaryDirectory() as tmp:
                root = synth_scene.write_merl(tmp)
                cfg = make_config('brdf', data_root=root, n_rays_per_step='512')
                model = get_model_class('brdf')(cfg).to(dev)
                from nerfactor_amd.nerfactor.datasets import get_dataset_class
                ds = get_dataset_class('brdf_merl')(cfg, 'train', device=dev)
                batch = next(iter(ds.build_pipeline(no_batch=True, seed=1)))
            bs = 512
        else:
            cfg = make_config(name, **extra)
            model = get_model_class(name)(cfg).to(dev)
            xyz = t(rng.uniform(-1, 1, size=(n, 3)))
            cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
            if name == 'nerf':
                batch = (None, None, cam, xyz - cam, t(rng.uniform(size=(n, 3))))
            else:
                batch = (None, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))),
                         mark_all_foreground(torch.ones(n, 1, device=dev)), xyz,
                         torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1), t(rng.uniform(size=(n, 512))))
            bs = n
        opt = optim.make_optimizer(model, cfg)
        loss, _ = optim.train_step(model, batch, opt, bs)
        out['%s_%d_lds%s' % (name, n, lds)] = (opt.bucket.flat.clone().cpu(), float(loss))
if mode == 'save':
    torch.save(out, path)
    print('saved', {k: v[1] for k, v in out.items()})
else:
    ref = torch.load(path)
    for k in out:
        same = torch.equal(out[k][0], ref[k][0])
        d = float((out[k][0] - ref[k][0]).abs().max())
        print(k, 'bit-identical' if same else 'DIFFERENT max|d| %.3g (|g| max %.3g)' % (d, float(ref[k][0].abs().max())))
