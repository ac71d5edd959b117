#!/bin/bash
# Round 3, call N: what bounds the NeRF ring backward — fetch distance 2..5 and a timing-only build without the epilogue
# stores, forward + backward only (no optimizer step: the experiment builds must not change the weights)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03n
mkdir -p $OUT
for name in ${NAMES:-prod nring_d2 nring_d3 nring_d4 nring_nost prod}; do
  lib=$PWD/nerfactor_amd/libnfx_$name.so; [ $name = prod ] && lib=$PWD/nerfactor_amd/libnfx.so
  (cd /tmp && NFX_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$name -o p -- \
     python $OLDPWD/scripts/bench_train.py --model nerf --steps 20 --warmup 3 --no-update > $OUT/run_$name.log 2>&1)
  f=$(find $OUT/prof_$name -name "*kernel_stats.csv" | head -1)
  cp "$f" $OUT/kernel_stats_$name.csv; rm -rf $OUT/prof_$name
  python - "$OUT/kernel_stats_$name.csv" "$name" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'nerf_bwd' in r['Name']:
        print('== %-11s %s calls %s avg %.1f us min %.1f max %.1f' % (sys.argv[2], r['Name'][10:34], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
done
