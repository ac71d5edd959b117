#!/bin/bash
# First GPU call of round 3 (≈8 min): what round 2 wrote after its GPU budget was spent, checked on hardware.
#   1. the hipGraph capture fed from the dataset loaders (optim._detached: r02 crashed in capture_end)
#      — each configuration in its own process under `timeout`, so a crash costs one line, not the call;
#   2. the loader benchmark, all four configurations;
#   3. the labels of `bench.py --precision fp32` (a short run);
#   4. the experiment variants written blind at the end of round 2 — NeRF MLP 9 (loop-top work under earlier tiles'
#      MFMAs) and 10 (32-bit point indices), light visibility 9 (32-bit row indices): bit-identity against the
#      defaults, then alternating A/B runs.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${TAG:-r03first}
mkdir -p $OUT
for cfg in hipGraph:0 hipGraph:2; do
  timeout 60 python -X faulthandler scripts/bench_loader.py --imh 128 --views 12 --epochs 2 --only $cfg \
      > $OUT/loader_$cfg.json 2> $OUT/loader_$cfg.err
  echo "$cfg rc=$? $(cut -c1-300 $OUT/loader_$cfg.json)"; grep -c "AccumulateGrad" $OUT/loader_$cfg.err
done
timeout 120 python scripts/bench_loader.py > $OUT/bench_loader.json 2> $OUT/bench_loader.err; echo "loader rc=$?"; cut -c1-900 $OUT/bench_loader.json
# 4. variant 9 of the NeRF MLP kernel (nerf_mlp_v9.hip, written blind): bit-identity against the default, then A/B
timeout 120 python - > $OUT/v9_check.log 2>&1 <<PY
import os, numpy as np, torch
from nerfactor_amd import build; build.build()
from nerfactor_amd import ops, synth
nets = synth.nerf_nets(seed=0)
blob = ops.pack_nerf_weights(*synth.nerf_layers(nets[1])).cuda()
rng = np.random.default_rng(0)
for n_rays, s in ((4096, 192), (301, 5), (70000, 64)):
    o = torch.from_numpy(rng.uniform(-1, 1, (n_rays, 3)).astype(np.float32)).cuda()
    d = torch.nn.functional.normalize(torch.from_numpy(rng.normal(size=(n_rays, 3)).astype(np.float32)), dim=1).cuda()
    z = torch.sort(torch.from_numpy(rng.uniform(2, 6, (n_rays, s)).astype(np.float32)), 1)[0].cuda()
    os.environ["NFX_NERF_VARIANT"] = "7"; a = ops.nerf_mlp_fwd(o, d, z, blob)
    for v in ("9", "10"):
        os.environ["NFX_NERF_VARIANT"] = v; b = ops.nerf_mlp_fwd(o, d, z, blob)
        print("nerf variant", v, n_rays, s, "bit-identical" if torch.equal(a, b) else "DIFFERENT max %g" % float((a - b).abs().max()))
os.environ["NFX_NERF_VARIANT"] = "7"
# light visibility: variant 9 (32-bit row indices) against the default 8
from nerfactor_amd import _capi
from oracle import nerfactor_ref as R
layers, out = R.init_mlp128(rng, 90, 1)
lblob = ops.pack_mlp128_weights([k for k, _ in layers] + [out[0][0]], [b for _, b in layers] + [out[0][1]], _capi.IN_XYZ_LDIR, 1).cuda()
lxyz = torch.from_numpy(R.gen_light_xyz(16, 32)[0].reshape(-1, 3).astype(np.float32)).cuda()
for n in (70, 1031, 40000):
    xyz = torch.from_numpy(rng.uniform(-1, 1, (n, 3)).astype(np.float32)).cuda()
    os.environ["NFX_LVIS_VARIANT"] = "8"; a = ops.lvis_fwd(xyz, lxyz, lblob)
    os.environ["NFX_LVIS_VARIANT"] = "9"; b = ops.lvis_fwd(xyz, lxyz, lblob)
    print("lvis variant 9", n, "bit-identical" if torch.equal(a, b) else "DIFFERENT max %g" % float((a - b).abs().max()))
PY
cat $OUT/v9_check.log | tail -10
for v in 7 9 10 7 9 10; do
  NFX_NERF_VARIANT=$v timeout 100 python bench.py --legs nerf --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_v$v.json 2>/dev/null
  python -c "import json;j=json.load(open('$OUT/bench_v$v.json'));print('variant $v', j['value'], j['roofline']['achieved'])"
done
for v in 8 9 8 9; do
  NFX_LVIS_VARIANT=$v timeout 100 python bench.py --legs nerfactor_microfacet --steps 6 --warmup 2 --no-cpu-baseline > $OUT/bench_lvis$v.json 2>/dev/null
  python -c "import json;j=json.load(open('$OUT/bench_lvis$v.json'))['nerfactor']['nerfactor_microfacet'];print('lvis variant $v', j['ms_per_step'], j['roofline']['avg_launch_ms'], j['roofline']['achieved'])"
done
timeout 200 python bench.py --steps 3 --warmup 1 --precision fp32 --cpu-budget 6 > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err
echo "bench fp32 rc=$?"; python - <<PY
import json
j = json.load(open("$OUT/bench_fp32.json"))
print(j["dtype"], j["value"], j["roofline"]["kernel"])
for k, v in j["nerfactor"].items():
    print(k, v["ms_per_step"], v["roofline"]["kernel"])
PY
