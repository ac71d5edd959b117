#!/bin/bash
# Round 3, call O: PMC counters of the NeRF ring backward (forward + backward only, no optimizer step) — where the
# activation stores are held up
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$PWD/gpurun_out/r03o
mkdir -p $OUT/pmc_csv
pass() {
  name=$1; shift
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc/$name -o p -- \
     python $R/scripts/bench_train.py --model ${MODEL:-nerf} --steps 3 --warmup 1 --no-update > $OUT/pmc_$name.log 2>&1)
  f=$(find $OUT/pmc/$name -name "*counter_collection.csv" | head -1); mkdir -p $OUT/pmc_csv/$name
  cp "$f" $OUT/pmc_csv/$name/p_counter_collection.csv 2>/dev/null; echo "pmc $name: $(wc -l < $OUT/pmc_csv/$name/p_counter_collection.csv) rows"
  rm -rf $OUT/pmc/$name
}
pass wave SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
pass mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD GRBM_GUI_ACTIVE
pass fifo SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL GRBM_GUI_ACTIVE
pass ta TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp1 TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum GRBM_GUI_ACTIVE
pass tcp2 TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum
pass tcc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_EA0_WRREQ_64B_sum
pass hbm FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
python scripts/pmc_digest.py $OUT/pmc_csv > $OUT/pmc_digest.json
python - <<'PY'
import json
d = json.load(open('gpurun_out/r03o/pmc_digest.json'))
for k, v in d.items():
    if 'bwd' in k or 'wgrad_lds' in k:
        print(k)
        for c, x in sorted(v.items()):
            print('   %-44s %.4g' % (c, x))
PY
