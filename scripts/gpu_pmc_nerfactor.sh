#!/bin/bash
# PMC passes for the NeRFactor render (one counter group per pass; never combined with other tracing).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/pmc_nerfactor
mkdir -p $OUT
cd /tmp
run_pass() {
  local name=$1; shift
  timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o p -- python $R/bench.py --legs ${LEGS:-nerfactor_microfacet,nerfactor} --steps 2 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
  echo "$name rc=$?"
}
run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS
run_pass fetch FETCH_SIZE
run_pass write WRITE_SIZE
run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
echo done
