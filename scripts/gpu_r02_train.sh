#!/bin/bash
# r02: GPU test suite after the training-side changes, then training-step benches + kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${TAG:-r02g}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x ${PYTEST_ARGS:-} > $OUT/pytest_full.log 2>&1; tail -25 $OUT/pytest_full.log
TAG=${TAG:-r02g} MODELS="${MODELS:-nerfactor_microfacet nerfactor nerf shape}" bash scripts/gpu_r02_prof_train.sh 2>&1 | tail -8
