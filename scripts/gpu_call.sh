#!/bin/bash
# One GPU-box call (via gpurun): runs the stages named on the command line, everything worth keeping goes to gpurun_out/<tag>/.
#   bash scripts/gpu_call.sh <tag> stage [stage ...]
# stages: pytest | pytest-x | smoke | bench | bench-norefine | soak | soak-xp | sigma | prof | pmc | train-prof
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
ROOT=$PWD
{ rocm-smi --showproductname 2>/dev/null | head -8; nproc; lscpu | grep -E "Model name|^CPU\(s\)"; } > $OUT/gpu.txt
for st in "$@"; do
  echo "=== $st $(date +%T)"
  case $st in
    pytest)   timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -25 $OUT/pytest_gpu.log ;;
    pytest-x) timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu_x.log 2>&1; tail -8 $OUT/pytest_gpu_x.log ;;
    smoke)    timeout 600 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; tail -6 $OUT/smoke.log ;;
    bench)    timeout 1200 python bench.py --steps ${STEPS:-10} --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    bench-norefine) timeout 600 python bench.py --steps ${STEPS:-10} --warmup 3 --legs nerf --no-last-sample-refine > $OUT/bench_norefine.json 2> $OUT/bench_norefine.err; tail -c 1200 $OUT/bench_norefine.json ;;
    soak)     timeout 600 python scripts/soak_8wave.py > $OUT/soak_product.log 2>&1; tail -12 $OUT/soak_product.log ;;
    soak-xp)  NFX_LIB_PATH=$ROOT/nerfactor_amd/libnfx_xp.so timeout 600 python scripts/soak_8wave.py --geo0 > $OUT/soak_experiment_build.log 2>&1; tail -16 $OUT/soak_experiment_build.log ;;
    sigma)    timeout 600 python scripts/sigma_last_error.py > $OUT/sigma_last_error.json 2>&1; cat $OUT/sigma_last_error.json ;;
    prof)     (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs ${PROF_LEGS:-nerf,nerfactor_microfacet,nerfactor,olat} > $ROOT/$OUT/prof_run.log 2>&1); find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -r head -25 ;;
    train-prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_train -o train -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs train > $ROOT/$OUT/prof_train_run.log 2>&1); find $OUT/prof_train -name "*kernel_stats*" | head -1 | xargs -r head -30 ;;
    *) echo "unknown stage $st" ;;
  esac
done
echo done $(date +%T)
