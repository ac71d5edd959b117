#!/bin/bash
# Runs on the GPU box (via gpurun): GPU parity tests, bench for both kernel variants, rocprof stats.
# Everything worth keeping goes to gpurun_out/.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu.txt
nproc >> $OUT/gpu.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/gpu.txt
echo "=== pytest -m gpu" | tee $OUT/pytest_gpu.log
timeout 1500 python -m pytest tests -m gpu -q -k "${PYTEST_K:-test}" > $OUT/pytest_gpu_full.log 2>&1; tail -15 $OUT/pytest_gpu_full.log | tee -a $OUT/pytest_gpu.log
if [ "${SMOKE:-1}" = "1" ]; then
echo "=== smoke" | tee $OUT/smoke.log
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -5 | tee -a $OUT/smoke.log
fi
for v in ${VARIANTS-7 6}; do
  echo "=== bench variant $v" | tee $OUT/bench_v$v.log
  NFX_NERF_VARIANT=$v timeout 900 python bench.py --steps 5 --warmup 2 $( [ "$v" != "7" ] && echo --no-cpu-baseline ) 2>&1 | tail -3 | tee -a $OUT/bench_v$v.log
done
if [ "${PROFILE:-1}" = "1" ]; then
  echo "=== rocprofv3 kernel stats"
  (cd /tmp && NFX_NERF_VARIANT=${PROF_VARIANT:-7} timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$OUT/prof -o nerf -- python $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$OUT/prof_run.log 2>&1)
  ls -R $OUT/prof | head -20
  find $OUT/prof -name "*kernel_stats*" | head -2 | xargs -r head -20
  cat /sys/fs/cgroup/cpu.max 2>/dev/null >> $OUT/gpu.txt
fi
echo done
