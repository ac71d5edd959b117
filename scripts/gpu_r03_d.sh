#!/bin/bash
# Round 3, call D: (1) the graphed-step test with the fixed-point accumulators cleared by a kernel instead of hipMemsetAsync;
# (2) the rewritten OLAT kernel: parity test + the bench leg; (3) NeRF tile stamps, register-only tile (ablation 75), with
# the epilogue removed / redirected to scratch registers.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03d
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_train.py -x -q -k "graphed_train_step or bit_reproducible" > $OUT/pytest_graph.log 2>&1; echo "graph tests rc=$?"; grep -h "first differing\|passed\|failed" $OUT/pytest_graph.log | cut -c1-400
timeout 300 python -m pytest tests/test_gpu_nerfactor.py -x -q -k "olat or model_call" > $OUT/pytest_olat.log 2>&1; echo "olat tests rc=$?"; tail -2 $OUT/pytest_olat.log
timeout 300 python bench.py --legs olat --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_olat.json 2> $OUT/bench_olat.err; echo "bench olat rc=$?"
python -c "
import json; j = json.load(open('$OUT/bench_olat.json'))['olat']; print('olat', j['ms_per_step'], j['roofline'])"
for xp in 0 1 2; do
  for ab in 75 0; do
    NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_xp$xp.so NFX_ABLATE=$ab timeout 120 python scripts/v6_timing.py > $OUT/stamps_xp${xp}_ab$ab.log 2>&1
    echo "xp $xp ablate $ab: $(tail -2 $OUT/stamps_xp${xp}_ab$ab.log | tr '\n' ' ')"
  done
done
