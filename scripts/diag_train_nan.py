"""Loss curve and parameter statistics of a training run on one synthetic batch until check_numerics raises
(bench_train.py's setup: seed 5, data seed 100 + rank).   python scripts/diag_train_nan.py [model] [steps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from nerfactor_amd import optim  # noqa: E402
from nerfactor_amd.nerfactor.config import make_config  # noqa: E402
from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground  # noqa: E402
from nerfactor_amd.nerfactor.models import get_model_class  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'nerfactor_microfacet'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
anomaly_from = int(sys.argv[3]) if len(sys.argv) > 3 else 10 ** 9
dev = torch.device('cuda:0')
torch.manual_seed(5)
cfg = make_config(name, shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='')
model = get_model_class(name)(cfg).to(dev)
opt = optim.make_optimizer(model, cfg)
rng = np.random.default_rng(100)
n = 1024
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
xyz = t(rng.uniform(-1, 1, size=(n, 3)))
nrm = torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1)
cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
batch = (None, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))), mark_all_foreground(torch.ones(n, 1, device=dev)),
         xyz, nrm, t(rng.uniform(size=(n, 512))))
names = {id(p): k for k, p in model.named_parameters()}
losses = []
for i in range(steps):
    try:
        if i >= anomaly_from:
            with torch.autograd.detect_anomaly(check_nan=True):
                loss, to_vis = optim.train_step(model, batch, opt, n)
        else:
            loss, to_vis = optim.train_step(model, batch, opt, n)
        model.flush_numerics(block=True)
    except FloatingPointError as e:
        print("step %d: check_numerics raised: %s" % (i, e))
        break
    except RuntimeError as e:
        print("step %d: anomaly: %s" % (i, str(e)[:600]))
        for k in ('pred_rgb', 'pred_normal', 'pred_albedo', 'pred_brdf', 'pred_lvis'):
            v = to_vis.get(k)
            if v is not None:
                print("  previous step %s: min %.6g max %.6g" % (k, float(v.min()), float(v.max())))
        break
    losses.append(float(loss))
    if not np.isfinite(losses[-1]):
        print("step %d: non-finite loss" % i)
        break
print("losses[:5]", [round(x, 5) for x in losses[:5]], "last 25:", [round(x, 5) for x in losses[-25:]])
g = opt.bucket.flat[:-1]
off = 0
for p in opt.params:
    k = p.numel()
    gp, pp = g[off:off + k], p.detach().reshape(-1)
    bad = (~torch.isfinite(gp)).sum().item() + (~torch.isfinite(pp)).sum().item()
    if bad or float(pp.abs().max()) > 50 or float(gp.abs().max()) > 50:
        print("  %-28s |p|max %.3g |g|max %.3g non-finite %d" % (names.get(id(p), '?'), float(pp.abs().max()), float(gp.abs().max()), bad))
    off += k
with torch.no_grad():
    pred = model(batch, mode='vali')[0] if False else None
for k in ('pred_rgb', 'pred_normal', 'pred_albedo', 'pred_brdf', 'pred_lvis'):
    v = to_vis.get(k)
    if v is not None:
        print("  last good step %s: min %.4g max %.4g finite %s" % (k, float(v.min()), float(v.max()), bool(torch.isfinite(v).all())))
print("  light: min %.4g max %.4g" % (float(model.light.min()), float(model.light.max())))
