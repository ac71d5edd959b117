#!/bin/bash
# Round 3, call R: brdf_compact_kernel with the per-point frame parked in LDS against the round-2 form
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
for r in ${ROUNDS:-1 2}; do
  NFX_LIB_PATH=$R/nerfactor_amd/libnfx_prepark.so timeout 200 python scripts/brdf_ab.py save 2>&1 | grep -v amdgpu.ids | tail -2
  timeout 200 python scripts/brdf_ab.py check 2>&1 | grep -v amdgpu.ids | tail -4
done
NFX_LIB_PATH=$R/nerfactor_amd/libnfx_t.so timeout 200 python scripts/brdf_phases.py 2>&1 | tail -2
[ "${TESTS:-0}" = 1 ] && timeout 900 python -m pytest tests/test_gpu_nerfactor.py -x -q -m gpu 2>&1 | tail -3
