"""Timing-only ablation of a NeRF MLP kernel variant (needs a build with NFX_ABLATION_BUILD=1).
Variant 2 masks: 1 no weight DMA, 2 no barriers, 4 no MFMA, 8 no A ds_reads, 16 no epilogue, 32 no posenc.
Variant 5 masks (ABLATE_VARIANT=5): 1 no weight staging, 2 no barrier, 4 no MFMA, 8 no A ds_reads, 16 no epilogue,
64 no bias init."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nerfactor_amd import ops  # noqa: E402
from tests import common  # noqa: E402

dev = torch.device('cuda:0')
blob = ops.pack_nerf_weights(*common.nerf_layers(common.nerf_nets(seed=0)[0])).to(dev)
n, s = 640000, 192
o = torch.randn(n, 3, device=dev)
d = torch.nn.functional.normalize(torch.randn(n, 3, device=dev), dim=1)
z = torch.sort(torch.rand(n, s, device=dev) * 4 + 2, dim=1)[0]
VARIANT = os.environ.get('ABLATE_VARIANT', '2')
os.environ['NFX_NERF_VARIANT'] = VARIANT
res = {}
MASKS = [0, 1, 2, 3, 4, 8, 12, 16, 32, 28, 31] if VARIANT == '2' else [0, 1, 2, 3, 4, 8, 16, 64, 80, 83, 7, 12]
if VARIANT == '7':   # variant 7 (LDS-DMA): 1 no DMA / vmcnt wait, 2 no barrier, 4 no MFMA, 8 no A reads, 64 no bias reads
    MASKS = [0, 1, 2, 3, 4, 8, 64, 66, 75]
for mask in MASKS:
    os.environ['NFX_ABLATE'] = str(mask)
    ops.nerf_mlp_fwd(o, d, z, blob)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(2):
        ops.nerf_mlp_fwd(o, d, z, blob)
    ev[1].record()
    torch.cuda.synchronize()
    ms = ev[0].elapsed_time(ev[1]) / 2
    res[mask] = ms
    print("mask %2d: %8.2f ms  (%.0f TF-equivalent)" % (mask, ms, n * s * 2 * 593408 / ms / 1e9), flush=True)
os.environ['NFX_ABLATE'] = '0'
for v in ('1', '0'):
    os.environ['NFX_NERF_VARIANT'] = v
    ops.nerf_mlp_fwd(o, d, z, blob)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(2):
        ops.nerf_mlp_fwd(o, d, z, blob)
    ev[1].record()
    torch.cuda.synchronize()
    print("variant %s: %.2f ms" % (v, ev[0].elapsed_time(ev[1]) / 2), flush=True)
print(json.dumps(res))
