#!/bin/bash
# Experiment build of ONE source file: bash scripts/build_obj_variant.sh <file.hip> <name> <extra flags...>
#   -> nerfactor_amd/libnfx_<name>.so = the product objects of build/obj with this one file recompiled with the flags
# (use with NFX_LIB_PATH; the product library is not touched)
set -e
cd "$(dirname "$0")/.."
src=$1; name=$2; shift 2
base=$(basename $src)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -Wno-unused-result -Iinclude \
    "$@" -x hip -c nerfactor_amd/csrc/$base -o build/var_${name}.o
objs=$(ls build/obj/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs build/var_${name}.o -o nerfactor_amd/libnfx_$name.so
echo built nerfactor_amd/libnfx_$name.so
