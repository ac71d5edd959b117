#!/bin/bash
# Round 3, call F: which form of "healing" the packed-op-written B registers is cheapest (stamps of the default NeRF kernel)?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03f
mkdir -p $OUT
for xp in ${XPS:-64 2048}; do
  for ab in 75 0; do
    NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_xp$xp.so NFX_ABLATE=$ab timeout 120 python scripts/v6_timing.py > $OUT/stamps_xp${xp}_ab$ab.log 2>&1
    echo "xp $xp ablate $ab: $(tail -2 $OUT/stamps_xp${xp}_ab$ab.log | tr '\n' ' ')"
  done
done
