"""Where does the host time of a training step go?  Host wall time and kernel launches per phase (model call, loss,
backward, optimizer) of the launch-bound NeRFactor step, plus the aten ops that launch the most kernels."""
import collections, json, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from nerfactor_amd import optim
from nerfactor_amd.nerfactor.config import make_config
from nerfactor_amd.nerfactor.models import get_model_class
name = sys.argv[1] if len(sys.argv) > 1 else 'nerfactor_microfacet'
dev = torch.device('cuda:0')
torch.manual_seed(5)
cfg = make_config(name, **(dict(shape_mode='finetune', shape_model_ckpt='none', test_envmap_dir='') if 'nerfactor' in name else {}))
model = get_model_class(name)(cfg).to(dev)
opt = optim.make_optimizer(model, cfg)
rng = np.random.default_rng(100)
n = 1024
t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
xyz = t(rng.uniform(-1, 1, size=(n, 3)))
nrm = torch.nn.functional.normalize(t(rng.normal(size=(n, 3))), dim=1)
cam = t(np.broadcast_to([2.2, -2.4, 1.7], (n, 3)))
batch = (None, None, cam, t(np.zeros((n, 3))), t(rng.uniform(size=(n, 3))), torch.ones(n, 1, device=dev), xyz, nrm,
         t(rng.uniform(size=(n, 512))))
from nerfactor_amd.nerfactor.datasets.nerf_shape import mark_all_foreground
mark_all_foreground(batch[5])
if name == 'nerf':
    batch = (None, None, cam, xyz - cam, t(rng.uniform(size=(n, 3))))
phases = collections.OrderedDict((k, 0.) for k in ('flush+zero_grad', 'call', 'loss', 'backward', 'opt.step'))


def step(rec=None):
    import contextlib
    R = (lambda s: torch.profiler.record_function(s)) if rec else (lambda s: contextlib.nullcontext())
    t0 = time.perf_counter()
    with R('P:flush+zero_grad'):
        model.flush_numerics() if hasattr(model, 'flush_numerics') else None; opt.zero_grad()
    t1 = time.perf_counter()
    with R('P:call'):
        pred, gt, kw, _ = model(batch, mode='train')
    t2 = time.perf_counter()
    with R('P:loss'):
        kw['keep_batch'] = True
        w = model.compute_loss(pred, gt, **kw).sum() / n
    t3 = time.perf_counter()
    with R('P:backward'):
        w.backward()
    t4 = time.perf_counter()
    with R('P:opt.step'):
        opt.step(loss=w.detach())
    t5 = time.perf_counter()
    for k, a, b in zip(phases, (t0, t1, t2, t3, t4), (t1, t2, t3, t4, t5)):
        phases[k] += b - a


for _ in range(5):
    step()
torch.cuda.synchronize()
for k in phases:
    phases[k] = 0.
K = 40
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / K
print(name, 'ms per step %.3f; host ms per phase:' % (wall * 1e3), {k: round(v / K * 1e3, 3) for k, v in phases.items()})
with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU, torch.profiler.ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(3):
        step(rec=True)
    torch.cuda.synchronize()
ev = list(prof.events())
ranges = [(e.name[2:], e.time_range.start, e.time_range.end) for e in ev if e.name.startswith('P:')]
launch = [e for e in ev if 'LaunchKernel' in e.name or e.name in ('hipMemcpyAsync', 'hipMemsetAsync', 'hipMemcpyWithStream')]
count = collections.Counter()
for e in launch:
    for nm, a, b in ranges:
        if a <= e.time_range.start <= b:
            count[nm] += 1
            break
print('launches per step by phase:', {k: round(v / 3, 1) for k, v in count.items()}, 'total', round(len(launch) / 3, 1))
ops = collections.Counter()
for e in ev:
    if e.name.startswith('aten::') and e.cpu_children == [] or e.name.startswith('aten::') and all('Launch' in c.name for c in e.cpu_children):
        ops[e.name] += 1
print('leaf aten ops per step:', [(k, round(v / 3, 1)) for k, v in ops.most_common(25)])
for e in ev:
    if e.name in ('aten::item', 'aten::_local_scalar_dense', 'aten::nonzero') and e.stack:
        print('SYNC', e.name, [f for f in e.stack if 'nerfactor_amd' in f or 'scripts' in f][:4])
