#!/bin/bash
# Round 3, call J: rocprofv3 kernel stats of the training steps, [feature][row] (old) vs feature-pair-major (new) layout
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03j
mkdir -p $OUT
for l in old new; do
  if [ $l = old ]; then export NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_old.so; else unset NFX_LIB_PATH; fi
  for m in nerfactor_microfacet nerf; do
    (cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_${l}_$m -o p -- \
       python $OLDPWD/scripts/bench_train.py --model $m --steps 10 --warmup 3 > $OUT/run_${l}_$m.log 2>&1)
    f=$(find $OUT/prof_${l}_$m -name "*kernel_stats.csv" | head -1)
    cp "$f" $OUT/train_step_${m}_${l}_kernel_stats.csv 2>/dev/null
    rm -rf $OUT/prof_${l}_$m
    echo "== $l $m"; head -8 $OUT/train_step_${m}_${l}_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
  done
done
