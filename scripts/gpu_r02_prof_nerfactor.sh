#!/bin/bash
# r02: rocprofv3 kernel trace of the NeRFactor render legs (where do the non-MLP milliseconds go?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r02c
mkdir -p $OUT
for leg in nerfactor_microfacet nerfactor; do
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$leg -o p -- \
     python $OLDPWD/bench.py --steps 4 --warmup 1 --legs $leg --no-cpu-baseline > $OUT/run_$leg.log 2>&1)
  f=$(find $OUT/prof_$leg -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/${leg}_kernel_stats.csv 2>/dev/null
  head -25 $OUT/${leg}_kernel_stats.csv | cut -c1-200
  rm -rf $OUT/prof_$leg
done
