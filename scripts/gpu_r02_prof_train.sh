#!/bin/bash
# r02: rocprofv3 kernel trace of the training steps (nerfactor_microfacet, nerf), 1024 rays per step
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/${TAG:-r02f}
mkdir -p $OUT
for m in ${MODELS:-nerfactor_microfacet nerf}; do
  timeout 300 python scripts/bench_train.py --model $m --steps 20 | tail -1 | tee $OUT/bench_train_$m.json
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$m -o p -- \
     python $OLDPWD/scripts/bench_train.py --model $m --steps 10 --warmup 3 > $OUT/run_$m.log 2>&1)
  f=$(find $OUT/prof_$m -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/train_step_${m}_kernel_stats.csv 2>/dev/null
  rm -rf $OUT/prof_$m
done
