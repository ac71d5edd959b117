#!/bin/bash
# Round 3, call P: non-temporal activation stores in the NeRF ring backward (timing + the TCP request latencies)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$PWD/gpurun_out/r03p
mkdir -p $OUT/pmc_csv
NAMES="prod nring_nt prod nring_nt" bash scripts/gpu_r03_n.sh
for name in prod nring_nt; do
  lib=$R/nerfactor_amd/libnfx_$name.so; [ $name = prod ] && lib=$R/nerfactor_amd/libnfx.so
  (cd /tmp && NFX_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace --pmc TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $OUT/pmc/$name -o p -- \
     python $R/scripts/bench_train.py --model nerf --steps 3 --warmup 1 --no-update > $OUT/pmc_$name.log 2>&1)
  f=$(find $OUT/pmc/$name -name "*counter_collection.csv" | head -1); mkdir -p $OUT/pmc_csv/$name
  cp "$f" $OUT/pmc_csv/$name/p_counter_collection.csv 2>/dev/null; rm -rf $OUT/pmc/$name
done
python scripts/pmc_digest.py $OUT/pmc_csv > $OUT/pmc_digest.json
python - <<'PY'
import csv, collections
for name in ('prod', 'nring_nt'):
    per = collections.defaultdict(float); n = set()
    for r in csv.DictReader(open('gpurun_out/r03p/pmc_csv/%s/p_counter_collection.csv' % name)):
        if 'nerf_bwd' in r['Kernel_Name']:
            per[r['Counter_Name']] += float(r['Counter_Value']); n.add(r['Dispatch_Id'])
    print(name, 'read latency %.0f cycles  write latency %.0f cycles' % (
        per['TCP_TCC_READ_REQ_LATENCY_sum'] / per['TCP_TCC_READ_REQ_sum'], per['TCP_TCC_WRITE_REQ_LATENCY_sum'] / per['TCP_TCC_WRITE_REQ_sum']))
PY
