#!/bin/bash
# Round 3, call Q: non-temporal activation stores (default) against plain stores (libnfx_nont.so): gradient identity,
# step times of all models, kernel stats of the nerf / nerfactor_microfacet steps
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$PWD/gpurun_out/r03q
mkdir -p $OUT
NFX_LIB_PATH=$R/nerfactor_amd/libnfx_nont.so NFX_NERF_BWD=0 NFX_M128_BWD=0 timeout 300 python scripts/grad_identity.py save 2>&1 | tail -1 | cut -c1-200
timeout 300 python scripts/grad_identity.py check 2>&1 | tail -10
for r in 1 2; do
  for v in nont prod; do
    lib=$R/nerfactor_amd/libnfx_$v.so; [ $v = prod ] && lib=$R/nerfactor_amd/libnfx.so
    for m in nerf nerfactor_microfacet nerfactor shape; do
      NFX_LIB_PATH=$lib timeout 120 python scripts/bench_train.py --model $m --steps 60 2>/dev/null | tail -1 | python -c "import sys, json
j = json.loads(sys.stdin.read()); print('$v train $m %.3f ms/step loss %.6f' % (j['ms_per_step'], j['final_loss']))"
    done
  done
done
for v in nont prod; do
  lib=$R/nerfactor_amd/libnfx_$v.so; [ $v = prod ] && lib=$R/nerfactor_amd/libnfx.so
  for m in nerf nerfactor_microfacet; do
    (cd /tmp && NFX_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- \
       python $R/scripts/bench_train.py --model $m --steps 10 --warmup 3 > $OUT/run_${v}_$m.log 2>&1)
    f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/train_step_${m}_$v.csv; rm -rf $OUT/prof
    python - $OUT/train_step_${m}_$v.csv "$v $m" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:5]:
    print('== %-28s %-44s calls %3s avg %7.1f us' % (sys.argv[2], r['Name'][:44], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  done
done
