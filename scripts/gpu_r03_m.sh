#!/bin/bash
# Round 3, call M: the NeRF ring backward (default) against the register-staged one (NFX_NERF_BWD=0) in one library:
# gradient identity, step times, rocprofv3 kernel stats, the NeRF training tests
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03m
mkdir -p $OUT
NFX_NERF_BWD=0 timeout 300 python scripts/grad_identity.py save 2>&1 | tail -1 | cut -c1-300
timeout 300 python scripts/grad_identity.py check 2>&1 | tail -12
for r in 1 2; do
  for v in 0 1; do
    NFX_NERF_BWD=$v timeout 120 python scripts/bench_train.py --model nerf --steps 60 2>/dev/null | tail -1 | python -c "import sys, json
j = json.loads(sys.stdin.read()); print('NFX_NERF_BWD=$v train nerf %.3f ms/step loss %.6f' % (j['ms_per_step'], j['final_loss']))"
  done
done
for v in 0 1; do
  (cd /tmp && NFX_NERF_BWD=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o p -- \
     python $OLDPWD/scripts/bench_train.py --model nerf --steps 10 --warmup 3 > $OUT/run_$v.log 2>&1)
  f=$(find $OUT/prof_$v -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/train_step_nerf_bwd$v.csv; rm -rf $OUT/prof_$v
  echo "== NFX_NERF_BWD=$v"; head -6 $OUT/train_step_nerf_bwd$v.csv | cut -d, -f1-4 | cut -c1-120
done
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_grads.py -x -q -m gpu 2>&1 | tail -5
