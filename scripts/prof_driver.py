"""One warm-up + N timed renders of the bench workload, nothing else (for rocprofv3 passes)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from nerfactor_amd import ops  # noqa: E402
from tests import common  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = torch.device('cuda:0')
nets, rayo, rayd = bench.synth_inputs(0)
blobs = [ops.pack_nerf_weights(*common.nerf_layers(n)).to(dev) for n in nets]
o, d = torch.from_numpy(rayo).to(dev), torch.from_numpy(rayd).to(dev)
for _ in range(1 + steps):
    bench.render_step(ops, o, d, blobs)
torch.cuda.synchronize()
print("ok")
