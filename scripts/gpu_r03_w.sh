#!/bin/bash
# Round 3, call W: matrix-pipe utilisation of the shipped learned-BRDF kernel (8 waves x 2 column tiles) — one PMC pass
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/r03w
mkdir -p $OUT/pmc_csv/sq
(cd /tmp && timeout 120 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs nerfactor > $OUT/pmc.log 2>&1)
cp "$(find $OUT/pmc -name '*counter_collection.csv' | head -1)" $OUT/pmc_csv/sq/p_counter_collection.csv; rm -rf $OUT/pmc
python scripts/pmc_digest.py $OUT/pmc_csv > $OUT/pmc_digest_brdf_nw8.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03w/pmc_digest_brdf_nw8.json'))
for k, v in d.items():
    if 'brdf_compact' in k or 'resident128' in k:
        print(k[:60], {c: round(x, 4) for c, x in v.items() if 'util' in c or 'frac' in c or c == 'dispatches'})
PY
