import sys, numpy as np, torch
sys.path.insert(0, '.')
from nerfactor_amd import ops
x = torch.linspace(-8.5, 8.5, 2000001, device='cuda')
for which, f in ((2, np.sin), (3, np.cos), (0, np.sin), (1, np.cos)):
    got = ops.selftest_sincos(x, which).cpu().numpy().astype(np.float64)
    want = f(x.cpu().numpy().astype(np.float64))
    print(which, 'max abs err', np.abs(got - want).max())
x = torch.linspace(-3.3, 3.3, 2000001, device='cuda')
for which, f in ((2, np.sin), (3, np.cos)):
    got = ops.selftest_sincos(x, which).cpu().numpy().astype(np.float64)
    print('small range', which, np.abs(got - f(x.cpu().numpy().astype(np.float64))).max())
