#!/bin/bash
# Instruction-cache counters of the NeRF render kernels (variant from NFX_NERF_VARIANT); names differ between ROCm
# releases, so the pass uses whatever `rocprofv3 -L` lists.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_icache_v${NFX_NERF_VARIANT:-5}
mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -oiE "\b(SQC?_[A-Z_]*(ICACHE|IFETCH|INST_CACHE)[A-Z_]*)\b" | sort -u > $OUT/icache_counters.txt
cat $OUT/icache_counters.txt
run_pass() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/scripts/prof_driver.py 1 > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_IFETCH_LEVEL"; do
  names=""
  for c in $grp; do grep -qx "$c" $OUT/icache_counters.txt && names="$names $c"; case $c in SQ_WAVE_CYCLES|SQ_WAIT_INST_ANY|SQ_BUSY_CYCLES) names="$names $c";; esac; done
  names=$(echo $names | tr ' ' '\n' | sort -u | tr '\n' ' ')
  [ -n "$names" ] && run_pass g$i $names
  i=$((i+1))
done
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get('OUTDIR', '')
for f in sorted(glob.glob('/root/repo/gpurun_out/pmc_icache_v*/g*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for row in csv.DictReader(open(f)):
        k = row['Kernel_Name'][:60]
        acc[k][row['Counter_Name']] += float(row['Counter_Value'])
    for k, v in acc.items():
        if 'nerf_mlp' in k or 'resident' in k:
            print(os.path.basename(os.path.dirname(os.path.dirname(f))), k, dict(v))
PY
echo done
