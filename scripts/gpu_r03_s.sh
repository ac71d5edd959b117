#!/bin/bash
# Round 3, call S: NeRF ring backward with 8 waves (256-row tiles, default) against 4 waves (NFX_NERF_BWD_NW=4) and the
# register-staged kernel (NFX_NERF_BWD=0): gradient identity, step times, kernel stats
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$PWD/gpurun_out/r03s
mkdir -p $OUT
NFX_NERF_BWD=0 timeout 300 python scripts/grad_identity.py save 2>&1 | tail -1 | cut -c1-120
echo "-- 8 waves"; timeout 300 python scripts/grad_identity.py check 2>&1 | grep nerf
echo "-- 4 waves"; NFX_NERF_BWD_NW=4 timeout 300 python scripts/grad_identity.py check 2>&1 | grep nerf
for r in 1 2; do
  for nw in 4 8; do
    NFX_NERF_BWD_NW=$nw timeout 120 python scripts/bench_train.py --model nerf --steps 60 2>/dev/null | tail -1 | python -c "import sys, json
j = json.loads(sys.stdin.read()); print('NW=$nw train nerf %.3f ms/step loss %.6f' % (j['ms_per_step'], j['final_loss']))"
  done
done
for nw in 4 8; do
  (cd /tmp && NFX_NERF_BWD_NW=$nw timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- \
     python $R/scripts/bench_train.py --model nerf --steps 10 --warmup 3 > $OUT/run_nw$nw.log 2>&1)
  f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/train_step_nerf_nw$nw.csv; rm -rf $OUT/prof
  python - $OUT/train_step_nerf_nw$nw.csv "NW=$nw" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:4]:
    print('== %-6s %-44s calls %3s avg %7.1f us min %7.1f max %7.1f' % (sys.argv[2], r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e3, float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
done
timeout 600 python -m pytest tests/test_gpu_train.py tests/test_gpu_reference_grads.py -x -q -m gpu 2>&1 | tail -2
