#!/bin/bash
# Round 3, call L: the ring backward (default) against the streamed one (NFX_M128_BWD=0) in one library: gradient identity,
# step times, rocprofv3 kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03l
mkdir -p $OUT
NFX_M128_BWD=0 timeout 300 python scripts/grad_identity.py save 2>&1 | tail -1 | cut -c1-300
timeout 300 python scripts/grad_identity.py check 2>&1 | tail -12
for r in 1 2; do
  for v in 0 1; do
    for m in nerfactor_microfacet nerfactor shape; do
      NFX_M128_BWD=$v timeout 120 python scripts/bench_train.py --model $m --steps 60 2>/dev/null | tail -1 | python -c "import sys, json
j = json.loads(sys.stdin.read()); print('NFX_M128_BWD=$v train $m %.3f ms/step loss %.6f' % (j['ms_per_step'], j['final_loss']))"
    done
  done
done
for v in 0 1; do
  (cd /tmp && NFX_M128_BWD=$v timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$v -o p -- \
     python $OLDPWD/scripts/bench_train.py --model nerfactor_microfacet --steps 10 --warmup 3 > $OUT/run_$v.log 2>&1)
  f=$(find $OUT/prof_$v -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/train_step_microfacet_bwd$v.csv; rm -rf $OUT/prof_$v
  echo "== NFX_M128_BWD=$v"; head -5 $OUT/train_step_microfacet_bwd$v.csv | cut -d, -f1-4 | cut -c1-120
done
