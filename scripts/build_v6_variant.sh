#!/bin/bash
# Product-mode experiment builds of nerf_mlp_v6.hip (no stamps): bash scripts/build_v6_variant.sh <name> <extra flags...>
#   -> nerfactor_amd/libnfx_<name>.so = the product objects of build/obj with this one file recompiled
set -e
cd "$(dirname "$0")/.."
name=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -Iinclude \
    -mllvm -amdgpu-mfma-vgpr-form "$@" -x hip -c nerfactor_amd/csrc/nerf_mlp_v6.hip -o build/v6_$name.o
objs=$(ls build/obj/*.o | grep -v nerf_mlp_v6.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/v6_$name.o -o nerfactor_amd/libnfx_$name.so
echo built nerfactor_amd/libnfx_$name.so
