#!/bin/bash
# Round 3, call T: timing-only — the wide weight-gradient GEMM reading tile-major addresses (NeRF step, no optimizer step)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$PWD/gpurun_out/r03t
mkdir -p $OUT
for name in ${NAMES:-prod wg_tilemajor prod wg_tilemajor}; do
  lib=$R/nerfactor_amd/libnfx_$name.so; [ $name = prod ] && lib=$R/nerfactor_amd/libnfx.so
  (cd /tmp && NFX_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- \
     python $R/scripts/bench_train.py --model ${MODEL:-nerf} --steps 20 --warmup 3 --no-update > $OUT/run_$name.log 2>&1)
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/kernel_stats_$name.csv; rm -rf $OUT/prof
  python - $OUT/kernel_stats_$name.csv "$name" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'wgrad_lds' in r['Name'] or 'bwd' in r['Name']:
        print('== %-14s %-40s calls %3s avg %7.1f us min %7.1f max %7.1f' % (sys.argv[2], r['Name'][:40], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
done
