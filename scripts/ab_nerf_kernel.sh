#!/bin/bash
# A/B of NeRF-kernel builds: bash scripts/ab_nerf_kernel.sh <lib suffix A> <lib suffix B> ...   (libnfx_<suffix>.so; "product" = libnfx.so)
# Prints min-of-5 kernel times, alternating twice, and whether every build's output equals the first one's bit for bit.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/ab
for r in 1 2; do
  for l in "$@"; do
    if [ $l = product ]; then unset NFX_LIB_PATH; else export NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_$l.so; fi
    timeout 100 python scripts/time_nerf_kernel.py 2>/dev/null
  done
done
first=$1
for l in "$@"; do
  if [ $l = product ]; then unset NFX_LIB_PATH; else export NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_$l.so; fi
  timeout 100 python - <<PY 2>/dev/null
import numpy as np, torch
from nerfactor_amd import ops, synth
blob = ops.pack_nerf_weights(*synth.nerf_layers(synth.nerf_nets(seed=0)[1])).cuda()
rng = np.random.default_rng(0)
outs = []
for n, s in ((4096, 192), (301, 5), (70000, 64)):
    o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)).cuda()
    d = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)), dim=1).cuda()
    z = torch.sort(torch.from_numpy(rng.uniform(2, 6, (n, s)).astype(np.float32)), 1)[0].cuda()
    outs.append(ops.nerf_mlp_fwd(o, d, z, blob).cpu())
if "$l" == "$first":
    torch.save(outs, "gpurun_out/ab/ref.pt"); print("$l: reference saved")
else:
    ref = torch.load("gpurun_out/ab/ref.pt")
    print("$l vs $first bit-identical:", all(torch.equal(a, b) for a, b in zip(outs, ref)))
PY
done
