#!/bin/bash
# Experiment builds of the default NeRF kernel for the cycle-stamp script: nerf_mlp_v6.hip alone is recompiled with
# -DNFX_V6_TIMING -DNFX_ABLATION_BUILD -DNFX_V6_FEW -DNFX_V6_XP=<mask> and linked with the product objects of build/obj.
#   bash scripts/build_v6_xp.sh 0 1 2     ->  nerfactor_amd/libnfx_xp{0,1,2}.so
set -e
cd "$(dirname "$0")/.."
python -m nerfactor_amd.build > /dev/null
for xp in "$@"; do
  (
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -Iinclude \
      -mllvm -amdgpu-mfma-vgpr-form -DNFX_ABLATION_BUILD -DNFX_V6_FEW -DNFX_V6_TIMING -DNFX_V6_XP=$xp ${XP_EXTRA:-} \
      -x hip -c nerfactor_amd/csrc/nerf_mlp_v6.hip -o build/v6_xp$xp.o
  objs=$(ls build/obj/*.o | grep -v nerf_mlp_v6.hip.o)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/v6_xp$xp.o -o nerfactor_amd/libnfx_xp$xp.so
  echo built nerfactor_amd/libnfx_xp$xp.so
  ) &
done
wait
