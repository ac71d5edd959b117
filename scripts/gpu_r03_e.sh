#!/bin/bash
# Round 3, call E: NeRF tile stamps with the freshly converted B registers (i) sent through an LDS round trip, (ii) re-written by
# v_mov_b32 — does an LDS return as last writer avoid the first-tile-of-a-layer cost that call D tied to VALU-written B operands?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03e
mkdir -p $OUT
for xp in 8 32; do
  for ab in 75 0; do
    NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_xp$xp.so NFX_ABLATE=$ab timeout 120 python scripts/v6_timing.py > $OUT/stamps_xp${xp}_ab$ab.log 2>&1
    echo "xp $xp ablate $ab: $(tail -2 $OUT/stamps_xp${xp}_ab$ab.log | tr '\n' ' ')"
  done
done
# bit-identity of the LDS round trip against the product kernel
NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_xp8.so timeout 120 python - <<'PY' > $OUT/xp8_identity.log 2>&1
import numpy as np, torch
from nerfactor_amd import ops, synth
blob = ops.pack_nerf_weights(*synth.nerf_layers(synth.nerf_nets(seed=0)[1])).cuda()
rng = np.random.default_rng(0)
n, s = 5000, 192
o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)).cuda()
d = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)), dim=1).cuda()
z = torch.sort(torch.from_numpy(rng.uniform(2, 6, (n, s)).astype(np.float32)), 1)[0].cuda()
a = ops.nerf_mlp_fwd(o, d, z, blob)
torch.save(a.cpu(), 'gpurun_out/r03e/xp8_out.pt')
print('xp8 finite', bool(torch.isfinite(a).all()), float(a.abs().mean()))
PY
timeout 120 python - <<'PY' >> $OUT/xp8_identity.log 2>&1
import numpy as np, torch
from nerfactor_amd import ops, synth
blob = ops.pack_nerf_weights(*synth.nerf_layers(synth.nerf_nets(seed=0)[1])).cuda()
rng = np.random.default_rng(0)
n, s = 5000, 192
o = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)).cuda()
d = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(n, 3)).astype(np.float32)), dim=1).cuda()
z = torch.sort(torch.from_numpy(rng.uniform(2, 6, (n, s)).astype(np.float32)), 1)[0].cuda()
a = ops.nerf_mlp_fwd(o, d, z, blob).cpu()
b = torch.load('gpurun_out/r03e/xp8_out.pt')
print('product vs xp8 bit-identical:', torch.equal(a, b))
PY
cat $OUT/xp8_identity.log | grep -v amdgpu.ids
