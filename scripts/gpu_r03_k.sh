#!/bin/bash
# Round 3, call K: PMC view of the NeRFactor training step's two big kernels (what bounds mlp128_bwd_kernel<1>?)
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03k
mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ|TA|TCP|TCC|TD|GRBM)_[A-Z0-9_]+" | sort -u > $OUT/counters_available.txt
wc -l $OUT/counters_available.txt
run_pass() {
  local name=$1; shift
  timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/scripts/bench_train.py --model nerfactor_microfacet --steps 6 --warmup 3 > $OUT/$name.log 2>&1
  echo "$name rc=$?"
  f=$(find $OUT/$name -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in rows:
    k = r['Kernel_Name'][:50]
    if 'mlp128_bwd_kernel<1>' in r['Kernel_Name'] or 'wgrad_lds_narrow' in r['Kernel_Name'] or 'resident128' in r['Kernel_Name']:
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); cnt[(k, r['Counter_Name'])] += 1
for k in agg:
    print(k, {c: round(v / cnt[(k, c)]) for c, v in agg[k].items()})
PY
}
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS
run_pass sq2 SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL
