#!/bin/bash
# r03 evidence pass on one box: GPU test suite, smoke, the driver's bench command, rocprofv3 kernel stats of the same
# command, PMC passes (one counter group per pass, never combined with tracing domains) for HBM traffic / MFMA busy.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/${TAG:-r03final}
mkdir -p $OUT
rocm-smi --showproductname 2>/dev/null | head -8 > $OUT/gpu_box.txt; nproc >> $OUT/gpu_box.txt; lscpu | grep -E "Model name|^CPU\(s\)" >> $OUT/gpu_box.txt
if [ "${TESTS:-1}" = "1" ]; then
  timeout 1500 python -m pytest tests -m gpu -q > $OUT/pytest_gpu.log 2>&1; tail -6 $OUT/pytest_gpu.log
  timeout 300 python __graft_entry__.py smoke 2>&1 | tail -2 | tee $OUT/smoke.log
fi
if [ "${BENCH:-1}" = "1" ]; then
  timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; cut -c1-600 $OUT/bench.json
fi
# rocprofv3 kernel stats of the same command, the render legs and the training legs traced separately: the NeRF MLP and
# light-visibility kernels also run (on 1024 rays) inside the training steps, which would blur their per-launch averages
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --legs nerf,nerfactor_microfacet,nerfactor,olat > $OUT/prof_run.log 2>&1)
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_kernel_stats.csv; rm -rf $OUT/prof
head -8 $OUT/bench_kernel_stats.csv | cut -c1-150
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs train > $OUT/prof_train_run.log 2>&1)
cp $(find $OUT/prof -name "*kernel_stats.csv" | head -1) $OUT/bench_train_legs_kernel_stats.csv; rm -rf $OUT/prof
head -8 $OUT/bench_train_legs_kernel_stats.csv | cut -c1-150
if [ "${PMC:-1}" = "1" ]; then
  run_pass() {
    local name=$1; shift
    (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc/$name -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs nerf,nerfactor_microfacet,nerfactor,olat > $OUT/pmc_$name.log 2>&1)
    f=$(find $OUT/pmc/$name -name "*counter_collection.csv" | head -1); mkdir -p $OUT/pmc_csv/$name; cp "$f" $OUT/pmc_csv/$name/p_counter_collection.csv 2>/dev/null; echo "pmc $name: $(wc -l < $OUT/pmc_csv/$name/p_counter_collection.csv) rows"
    rm -rf $OUT/pmc/$name
  }
  run_pass sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS
  run_pass fetch FETCH_SIZE
  run_pass write WRITE_SIZE
  run_pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
  python scripts/pmc_digest.py $OUT/pmc_csv > $OUT/pmc_digest.json; head -c 1500 $OUT/pmc_digest.json
fi
echo done
