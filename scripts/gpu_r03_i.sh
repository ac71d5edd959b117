#!/bin/bash
# Round 3, call I: feature-PAIR-major activations between the backward kernels and the weight-gradient GEMMs
# (feat_store.hpp): gradients bit-identical to the [feature][row] form of the previous commit? step times?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03i
mkdir -p $OUT
NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_old.so timeout 300 python scripts/grad_identity.py save 2>&1 | tail -2 | cut -c1-400
timeout 300 python scripts/grad_identity.py check 2>&1 | tail -14
for r in 1 2; do
  for l in old new; do
    if [ $l = old ]; then export NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_old.so; else unset NFX_LIB_PATH; fi
    for m in nerfactor_microfacet nerfactor nerf shape; do
      timeout 120 python scripts/bench_train.py --model $m --steps 60 2>/dev/null | tail -1 | python -c "import sys, json
j = json.loads(sys.stdin.read()); print('$l train $m %.3f ms/step loss %.6f' % (j['ms_per_step'], j['final_loss']))"
    done
  done
done
unset NFX_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_grads.py -x -q > $OUT/pytest_train.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_train.log
