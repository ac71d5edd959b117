#!/bin/bash
# A/B of the secondary kernels: default build vs every MFMA file in VGPR form (nerfactor_amd/libnfx_vf.so)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out/vf
for lib in default vf; do
  if [ $lib = vf ]; then export NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_vf.so; else unset NFX_LIB_PATH; fi
  for m in nerf shape nerfactor_microfacet nerfactor; do
    echo "== $lib train $m"; timeout 300 python scripts/bench_train.py --model $m --steps 20 2>&1 | tail -1 | tee gpurun_out/vf/train_${m}_$lib.json | cut -c1-400
  done
  echo "== $lib geometry"; timeout 600 python scripts/bench_geometry.py 2>&1 | tail -1 | tee gpurun_out/vf/geometry_$lib.json | cut -c1-400
done
