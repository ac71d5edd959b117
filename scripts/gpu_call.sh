#!/bin/bash
# One GPU-box call (via gpurun): runs the stages named on the command line, everything worth keeping goes to gpurun_out/<tag>/.
#   bash scripts/gpu_call.sh <tag> stage [stage ...]
# stages: bench-driver | rccl | pytest | pytest-x | pytest-k (PYTEST_K=expr) | smoke | bench | bench-norefine | bench-train | bench-train-unfused |
#         bench-train-fp32 | soak | sigma | fitted | generic | prof | train-prof | train-prof-fp32 | pmc | pmc-train | pmc-sq2 |
#         bench-legs (LEGS=..., LEGS_TAG=...) | step-trace (TRACE_MODEL=...) | ubench-pair
#   round 6, second half (one option A / B'd per stage on one box: NFX_<OPTION>=0 / 1 through the binding):
#         bench-train-nerf-ab (nerf_bwd_rows) | wgrad-slabs-sweep (SLABS=...) | wgrad-rounds-ab | nerf-streams-ab | nerf-bwd-rows |
#         prof-train-nerf | prof-nerf-bwd-rows (NBR_KIND=fitted|glorot) | geometry-ab (sigma_grad_rows) | sigma-variant-ab |
#         soak-nerf-bwd | soak-sigma-v6 (SOAK_LAUNCHES=...)
#   A / B stages (experiment builds: NFX_EXTRA_DEFS=... python -m nerfactor_amd.build --out nerfactor_amd/libnfx_xpX.so):
#         fused-ab (AB_LIBS=...) | ring-ab (RING_LIBS=...) | generic-ab | splits-ab | generic-prof | generic-pmc
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
export NFX_CONVERGENCE_OUT=${NFX_CONVERGENCE_OUT:-$PWD/gpurun_out/$1/convergence.json}
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
    bench-driver) T0=$SECONDS; timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_line.json 2> $OUT/bench_line.err; echo "driver-command wall $((SECONDS-T0)) s"; cp bench_detail.json $OUT/ 2>/dev/null; wc -c $OUT/bench_line.json; cat $OUT/bench_line.json; tail -3 $OUT/bench_line.err; python -c "
import json; d=json.load(open('$OUT/bench_detail.json')); print('wall_s', d.get('wall_s'))" ;;
    coarse-probe2) timeout 900 python scripts/coarse_refine_probe${PROBE:-2}.py > $OUT/coarse_refine_probe2.json 2> $OUT/coarse_refine_probe2.err; python - <<PYEOF
import json
d=json.load(open("$OUT/coarse_refine_probe2.json"))
for w,v in d.items():
    print(w, v['rays'])
    for k,r in v['results'].items(): print('   %-44s' % k[:44], {a:(round(b,5) if isinstance(b,float) else b) for a,b in r.items()} if isinstance(r,dict) else [round(x,4) for x in r])
PYEOF
              tail -3 $OUT/coarse_refine_probe2.err ;;
    coarse-probe) timeout 900 python scripts/coarse_refine_probe${PROBE:-}.py > $OUT/coarse_refine_probe.json 2> $OUT/coarse_refine_probe.err; python - <<PYEOF
import json
d=json.load(open("$OUT/coarse_refine_probe.json"))
for w,v in d.items():
    print(w, v['rays'])
    for k,r in v['results'].items(): print('   %-44s' % k, {a:(round(b,5) if isinstance(b,float) else b) for a,b in r.items()})
PYEOF
              tail -3 $OUT/coarse_refine_probe.err ;;
    residual) timeout 600 python scripts/coarse_refine_residual.py > $OUT/coarse_refine_residual.json 2> $OUT/coarse_refine_residual.err; cat $OUT/coarse_refine_residual.json; tail -3 $OUT/coarse_refine_residual.err ;;
    soak-libs) for lib in ${SOAK_LIBS:-libnfx_xp libnfx_xpN libnfx_xpS}; do echo "--- $lib"; NFX_LIB_PATH=$ROOT/nerfactor_amd/$lib.so REPS=${REPS:-30} timeout 500 python scripts/soak_8wave.py --geo0 > $OUT/soak_$lib.log 2>&1; grep -v "^  launch" $OUT/soak_$lib.log | tail -${SOAK_TAIL:-14}; done ;;
    time-refine) timeout 600 python scripts/time_refine.py > $OUT/time_refine.json 2> $OUT/time_refine.err; cat $OUT/time_refine.json; tail -3 $OUT/time_refine.err ;;
    bench-traffic) timeout 1200 python bench.py --gpus 1 --steps 10 --warmup 3 --measure-traffic --legs nerf,nerfactor_microfacet,olat > $OUT/bench_line_measured_traffic.json 2> $OUT/bench_traffic.err; cp bench_detail.json $OUT/bench_detail_measured_traffic.json; python -c "
import json; d=json.load(open('$OUT/bench_detail_measured_traffic.json')); print('nerf', d['roofline']['traffic'], d['roofline']['traffic_source'][:60]); print('lvis', d['nerfactor']['nerfactor_microfacet']['roofline']['traffic'], d['nerfactor']['nerfactor_microfacet']['roofline']['traffic_source'][:40]); print('olat', d['olat']['roofline']['traffic'], d['olat']['roofline']['traffic_source'][:40])" ;;
    bench-nerfactor-ab) for r in 1 0 1 0; do NFX_LVIS_ROWS=$r timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --legs nerfactor_microfacet,nerfactor,olat --no-cpu-baseline > $OUT/bench_nerfactor_rows$r.json 2> $OUT/bench_nerfactor_rows$r.err; python -c "
import json; l=json.load(open('$OUT/bench_nerfactor_rows$r.json')); print('lvis_rows=$r', {k: round(v['ms_per_step'], 3) for k, v in l['legs'].items()})"; done ;;
    bench-nerfactor) timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --legs nerfactor_microfacet,nerfactor,olat,relight --no-cpu-baseline > $OUT/bench_nerfactor_line.json 2> $OUT/bench_nerfactor.err; python -c "
import json; l=json.load(open('$OUT/bench_nerfactor_line.json')); print(l['legs'])"; tail -3 $OUT/bench_nerfactor.err ;;
    rehearsal) NFX_BENCH_REHEARSAL=1 timeout 1500 python bench.py --gpus ${REH_GPUS:-2} --steps 2 --warmup 1 > $OUT/bench_rehearsal_line.json 2> $OUT/bench_rehearsal.err; echo rc=$?; cp bench_detail.json $OUT/bench_rehearsal_detail.json; head -c 1500 $OUT/bench_rehearsal_line.json; echo; tail -5 $OUT/bench_rehearsal.err ;;
    grad-modes) timeout 900 python scripts/grad_modes.py > $OUT/grad_modes.json 2> $OUT/grad_modes.err; python -c "
import json; d=json.load(open('$OUT/grad_modes.json'))
for m,v in d.items():
    for k,r in v.items(): print('%-22s %-32s worst %.4f median %.4f  loss1 %.1e' % (m,k,r['grad_rel_frobenius_vs_reference_worst'],r['grad_rel_frobenius_median_tensor'],r['loss_step1_rel_err']))"; tail -3 $OUT/grad_modes.err ;;
    bench-force-group) timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --force-group > $OUT/bench_force_group_line.json 2> $OUT/bench_force_group.err; echo rc=$?; python -c "
import json; l=json.loads([x for x in open('$OUT/bench_force_group_line.json') if x.startswith('{')][-1]); print(l['collective_backend'], l['value'], {k: v.get('ms_per_step') for k, v in l['legs'].items()})"; tail -3 $OUT/bench_force_group.err ;;
    rccl)     timeout 1200 python -m pytest tests/test_gpu_rccl.py -q -x > $OUT/pytest_rccl.log 2>&1; tail -30 $OUT/pytest_rccl.log ;;
    bench)    timeout 1200 python bench.py --steps ${STEPS:-10} --warmup 3 > $OUT/bench.json 2> $OUT/bench.err; tail -c 1500 $OUT/bench.json; tail -3 $OUT/bench.err ;;
    bench-legs) timeout 1200 python bench.py --steps ${STEPS:-5} --warmup 2 --legs ${LEGS:-geometry} ${BENCH_FLAGS:-} > $OUT/bench_${LEGS_TAG:-legs}.json 2> $OUT/bench_${LEGS_TAG:-legs}.err; tail -c ${TAILC:-3000} $OUT/bench_${LEGS_TAG:-legs}.json; tail -3 $OUT/bench_${LEGS_TAG:-legs}.err ;;
    ubench-pair) (cd scripts/ubench && timeout 600 ./two_wave_valu_pair.bin > $ROOT/$OUT/two_wave_valu_pair.jsonl 2>&1); cat $OUT/two_wave_valu_pair.jsonl ;;
    bench-norefine) timeout 600 python bench.py --steps ${STEPS:-10} --warmup 3 --legs nerf --no-last-sample-refine > $OUT/bench_norefine.json 2> $OUT/bench_norefine.err; tail -c 1200 $OUT/bench_norefine.json ;;
    soak)     timeout 600 python scripts/soak_8wave.py > $OUT/soak_product.log 2>&1; tail -12 $OUT/soak_product.log ;;
    pmc)      for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
                 set -- $pass; name=$1; shift
                 (cd /tmp && timeout 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/pmc/$name -o p -- python $ROOT/bench.py --legs ${PMC_LEGS:-nerf,nerfactor_microfacet,nerfactor,olat} --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/pmc_$name.log 2>&1); echo "pmc pass $name rc=$?"
               done
               python scripts/pmc_digest.py $OUT/pmc > $OUT/pmc_digest.json; python -c "
import json; d=json.load(open('$OUT/pmc_digest.json'))
for k,v in d.items():
    if 'mfma_util' in v and v.get('mfma_util',0)>0.05 or 'shade' in k: print(k[:60], {a: (round(b,4) if isinstance(b,float) and b<10 else int(b)) for a,b in v.items() if a in ('dispatches','mfma_util','hbm_read_bytes_corrected','hbm_write_bytes')})" ;;
    pmc-train) for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_LDS" "fetch FETCH_SIZE" "write WRITE_SIZE"; do
                 set -- $pass; name=$1; shift
                 (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/pmc_train/$name -o p -- python $ROOT/bench.py --legs train --train-models ${PMC_MODELS:-nerfactor_microfacet,nerf} --steps ${PMC_STEPS:-1} --warmup 1 --no-cpu-baseline --no-hip-graph > $ROOT/$OUT/pmc_train_$name.log 2>&1); echo "pmc pass $name rc=$?"
               done
               python scripts/pmc_digest.py $OUT/pmc_train > $OUT/pmc_train_digest.json; python - <<PYEOF
import json
d=json.load(open("$OUT/pmc_train_digest.json"))
for k,v in d.items():
    if 'bwd' in k or 'wgrad' in k:
        print(k[:70], {a: (round(b,4) if isinstance(b,float) and b<10 else int(b)) for a,b in v.items() if a in ('dispatches','mfma_util','SQ_WAIT_INST_ANY_frac_of_wave_cycles','SQ_WAIT_INST_LDS_frac_of_wave_cycles','SQ_LDS_BANK_CONFLICT','SQ_LDS_IDX_ACTIVE','hbm_read_bytes_corrected','hbm_write_bytes','SQ_BUSY_CYCLES','GRBM_GUI_ACTIVE','SQ_INSTS_LDS')})
PYEOF
               ;;
    soak-xp2) NFX_LIB_PATH=$ROOT/nerfactor_amd/libnfx_xp2.so timeout 600 python scripts/soak_8wave.py --geo0 > $OUT/soak_experiment_fastdiv.log 2>&1; tail -8 $OUT/soak_experiment_fastdiv.log ;;
    fitted)   timeout 600 python scripts/fitted_outliers.py > $OUT/fitted_outliers.json 2>&1; cat $OUT/fitted_outliers.json ;;
    pytest-k) timeout 1500 python -m pytest tests -m gpu -q ${PYTEST_FLAGS:-} -k "$PYTEST_K" > $OUT/pytest_gpu_k.log 2>&1; tail -${TAILN:-30} $OUT/pytest_gpu_k.log ;;
    bench-train) timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 --legs train --no-cpu-baseline > $OUT/bench_train.json 2> $OUT/bench_train.err; tail -c 2500 $OUT/bench_train.json; tail -3 $OUT/bench_train.err ;;
    bench-train-nerf-ab) for i in 1 2; do for r in 0 1; do NFX_NERF_BWD_ROWS=$r timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --legs train --train-models nerf --no-cpu-baseline > $OUT/bench_train_nerf_rows$r.json 2> $OUT/bench_train_nerf_rows$r.err; python - <<PYEOF
import json
d = json.load(open("bench_detail.json"))["train"]["nerf"]
print("nerf_bwd_rows=$r", {k: d.get(k) for k in ("ms_per_step", "ms_per_step_eager", "points_with_gradient_frac")}, {k: d["roofline"].get(k) for k in ("frac", "largest_backward_call_ms", "backward_calls_ms_per_step")})
PYEOF
      tail -2 $OUT/bench_train_nerf_rows$r.err; done; done ;;
    prof-train-nerf) for r in 0 1; do mkdir -p $ROOT/$OUT/prof_train_nerf$r; (cd /tmp && export TMPDIR=/tmp && NFX_NERF_BWD_ROWS=$r timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_train_nerf$r -o t -- python $ROOT/bench.py --steps 3 --warmup 1 --legs train --train-models nerf --no-cpu-baseline --no-hip-graph > $ROOT/$OUT/prof_train_nerf$r.log 2>&1); f=$(find $OUT/prof_train_nerf$r -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/train_nerf_rows${r}_kernel_stats.csv; rm -rf $OUT/prof_train_nerf$r; head -16 $OUT/train_nerf_rows${r}_kernel_stats.csv | cut -c1-150,250-330; done ;;
    prof-nerf-bwd-rows) for r in 0 1; do mkdir -p $ROOT/$OUT/prof_nbr$r; (cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_nbr$r -o t -- python $ROOT/scripts/nerf_bwd_rows.py ${NBR_KIND:-fitted} $r > $ROOT/$OUT/prof_nbr$r.log 2>&1); f=$(find $OUT/prof_nbr$r -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/nerf_${NBR_KIND:-fitted}_rows${r}_kernel_stats.csv; rm -rf $OUT/prof_nbr$r; head -12 $OUT/nerf_${NBR_KIND:-fitted}_rows${r}_kernel_stats.csv | cut -c1-100,200-330; done ;;
    wgrad-slabs-sweep) for sl in ${SLABS:-0 27 36 54 72 96 128}; do NFX_WGRAD_SLABS=$sl timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --legs train --train-models nerf --no-cpu-baseline > $OUT/bench_train_nerf_slabs$sl.json 2> $OUT/bench_train_nerf_slabs$sl.err; python - <<PYEOF
import json
d = json.load(open("bench_detail.json"))["train"]["nerf"]
print("wgrad_slabs=$sl", {k: d.get(k) for k in ("ms_per_step", "ms_per_step_eager", "points_with_gradient_frac")}, {k: d["roofline"].get(k) for k in ("largest_backward_call_ms", "backward_calls_ms_per_step")})
PYEOF
      done ;;
    wgrad-rounds-ab) for i in 1 2; do for r in 0 1 2; do NFX_WGRAD_ROUNDS=$r timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --legs train --train-models nerf --no-cpu-baseline > $OUT/bench_train_nerf_rounds$r.json 2> $OUT/bench_train_nerf_rounds$r.err; python - <<PYEOF
import json
d = json.load(open("bench_detail.json"))["train"]["nerf"]
print("wgrad_rounds=$r", {k: d.get(k) for k in ("ms_per_step", "ms_per_step_eager", "points_with_gradient_frac")}, {k: d["roofline"].get(k) for k in ("frac", "largest_backward_call_ms", "backward_calls_ms_per_step")})
PYEOF
      done; done ;;
    nerf-streams-ab) for i in 1 2; do for r in 0 1; do NFX_NERF_BWD_SIDE_STREAMS=$r timeout 600 python bench.py --steps ${STEPS:-5} --warmup 2 --legs train --train-models nerf --no-cpu-baseline > $OUT/bench_train_nerf_streams$r.json 2> $OUT/bench_train_nerf_streams$r.err; python - <<PYEOF
import json
d = json.load(open("bench_detail.json"))["train"]["nerf"]
print("side_streams=$r", {k: d.get(k) for k in ("ms_per_step", "ms_per_step_eager", "points_with_gradient_frac", "step")}, {k: d["roofline"].get(k) for k in ("frac", "largest_backward_call_ms", "backward_calls_ms_per_step")})
PYEOF
      tail -2 $OUT/bench_train_nerf_streams$r.err; done; done ;;
    geometry-ab) for i in 1 2; do for r in 0 1; do NFX_SIGMA_GRAD_ROWS=$r timeout 600 python bench.py --steps 3 --warmup 1 --legs geometry --no-cpu-baseline > $OUT/bench_geometry_rows$r.json 2> $OUT/bench_geometry_rows$r.err; python - <<PYEOF
import json
d = json.load(open("bench_detail.json"))["geometry"]["depth_normal"]
print("sigma_grad_rows=$r", {k: d.get(k) for k in ("ms_per_view", "rays_per_s", "samples_with_density_frac")}, {k: d["roofline"].get(k) for k in ("frac", "kernels_ms_per_view", "share_of_stage")})
PYEOF
      tail -2 $OUT/bench_geometry_rows$r.err; done; done ;;
    soak-nerf-bwd) timeout ${SOAK_TIMEOUT:-900} python scripts/soak_nerf_bwd.py --launches ${SOAK_LAUNCHES:-3000} > $OUT/soak_nerf_bwd.log 2>&1; tail -5 $OUT/soak_nerf_bwd.log ;;
    sigma-variant-ab) for i in 1 2; do for r in 0 1; do NFX_SIGMA_VARIANT=$r timeout 600 python bench.py --steps 3 --warmup 1 --legs geometry --no-cpu-baseline > $OUT/bench_geometry_sv$r.json 2> $OUT/bench_geometry_sv$r.err; python - <<PYEOF
import json
g = json.load(open("bench_detail.json"))["geometry"]
d, l = g["depth_normal"], g["light_visibility"]
print("sigma_variant=$r", {k: d.get(k) for k in ("ms_per_view", "rays_per_s")}, {k: d["roofline"].get(k) for k in ("frac", "kernels_ms_per_view")}, "| shadow rays", {k: l.get(k) for k in ("ms", "pairs_per_s", "full_view_estimate_s")}, {k: l["roofline"].get(k) for k in ("frac", "kernels_ms")})
PYEOF
      tail -2 $OUT/bench_geometry_sv$r.err; done; done ;;
    soak-sigma-v6) timeout ${SOAK_TIMEOUT:-900} python scripts/soak_sigma_v6.py --launches ${SOAK_LAUNCHES:-2000} > $OUT/soak_sigma_v6.log 2>&1; tail -4 $OUT/soak_sigma_v6.log ;;
    nerf-bwd-rows) timeout 600 python scripts/nerf_bwd_rows.py > $OUT/nerf_bwd_rows.json 2> $OUT/nerf_bwd_rows.err; cat $OUT/nerf_bwd_rows.json; tail -3 $OUT/nerf_bwd_rows.err ;;
    bench-train-fp32) for m in nerfactor_microfacet nerfactor nerf shape; do for fm in "pairs" "pairs --graph" "native"; do timeout 300 python scripts/bench_train.py --model $m --precision fp32 --fp32-matrix $fm --steps 10 >> $OUT/bench_train_fp32.jsonl 2>> $OUT/bench_train_fp32.err; done; done; cat $OUT/bench_train_fp32.jsonl; tail -3 $OUT/bench_train_fp32.err ;;
    bench-train-unfused) NFX_WGRAD_FUSED=0 timeout 900 python bench.py --steps ${STEPS:-5} --warmup 2 --legs train --no-cpu-baseline --train-models nerfactor_microfacet > $OUT/bench_train_unfused.json 2> $OUT/bench_train_unfused.err; tail -c 1500 $OUT/bench_train_unfused.json ;;
    soak-xp)  NFX_LIB_PATH=$ROOT/nerfactor_amd/libnfx_xp.so timeout 600 python scripts/soak_8wave.py --geo0 > $OUT/soak_experiment_build.log 2>&1; tail -16 $OUT/soak_experiment_build.log ;;
    generic)  timeout 600 python scripts/generic_rates.py > $OUT/generic_rates.json 2> $OUT/generic_rates.err; cat $OUT/generic_rates.json; tail -3 $OUT/generic_rates.err ;;
    generic-prof) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_generic -o g -- python $ROOT/scripts/generic_rates.py > $ROOT/$OUT/prof_generic_run.log 2>&1); find $OUT/prof_generic -name "*kernel_stats*" | head -1 | xargs -r head -12 ;;
    generic-pmc) for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS" "sq2 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_INSTS_MFMA"; do
                 set -- $pass; name=$1; shift
                 (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/pmc_generic/$name -o p -- python $ROOT/scripts/generic_rates.py > $ROOT/$OUT/pmc_generic_$name.log 2>&1); echo "pmc pass $name rc=$?"
               done
               python scripts/pmc_digest.py $OUT/pmc_generic > $OUT/pmc_generic_digest.json; python - <<PYEOF
import json
d=json.load(open("$OUT/pmc_generic_digest.json"))
for k,v in d.items():
    if 'generic' in k: print(k[:60], {a: (round(b,4) if isinstance(b,float) and b<10 else int(b)) for a,b in v.items()})
PYEOF
               ;;
    sigma)    timeout 600 python scripts/sigma_last_error.py > $OUT/sigma_last_error.json 2>&1; cat $OUT/sigma_last_error.json ;;
    prof)     (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof -o bench -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs ${PROF_LEGS:-nerf,nerfactor_microfacet,nerfactor,olat} > $ROOT/$OUT/prof_run.log 2>&1); find $OUT/prof -name "*kernel_stats*" | head -1 | xargs -r head -25 ;;
    train-prof) (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_train -o train -- python $ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --legs train > $ROOT/$OUT/prof_train_run.log 2>&1); find $OUT/prof_train -name "*kernel_stats*" | head -1 | xargs -r head -30 ;;
    train-prof-fp32) (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/prof_train_fp32 -o t -- python $ROOT/scripts/bench_train.py --model ${FP32_MODEL:-nerfactor_microfacet} --precision fp32 --fp32-matrix pairs --steps 10 > $ROOT/$OUT/prof_train_fp32_run.log 2>&1); find $OUT/prof_train_fp32 -name "*kernel_stats*" | head -1 | xargs -r head -40 ;;
    generic-ab) for mp in 1 0; do timeout 300 python scripts/generic_rates.py --only ${GENERIC_ONLY:-surface_128x4_lvis_fp32,surface_128x4_lvis,nerf_enc_256x8_fp32} --option wgrad_map=$mp > $OUT/generic_rates_map$mp.json 2> $OUT/generic_rates_map$mp.err; cat $OUT/generic_rates_map$mp.json; tail -2 $OUT/generic_rates_map$mp.err
                 for pass in "sq SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_INST_LDS" "fetch FETCH_SIZE" "write WRITE_SIZE" "tcc TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
                   set -- $pass; name=$1; shift
                   (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/pmc_generic_map$mp/$name -o p -- python $ROOT/scripts/generic_rates.py --only surface_128x4_lvis_fp32 --option wgrad_map=$mp > $ROOT/$OUT/pmc_generic_map${mp}_$name.log 2>&1); echo "pmc pass $name rc=$?"
                 done
                 python scripts/pmc_digest.py $OUT/pmc_generic_map$mp > $OUT/pmc_generic_map${mp}_digest.json; python - <<PYEOF
import json
d=json.load(open("$OUT/pmc_generic_map${mp}_digest.json"))
for k,v in d.items():
    if 'generic' in k: print(k[:60], {a: (round(b,4) if isinstance(b,float) and b<10 else int(b)) for a,b in v.items()})
PYEOF
               done ;;
    pmc-sq2) for pass in "sq2 SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM" "sq3 SQ_WAVE_CYCLES SQ_INST_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES"; do
                 set -- $pass; name=$1; shift
                 (cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/pmc_sq2_train/$name -o p -- python $ROOT/bench.py --legs train --train-models nerfactor_microfacet --steps 1 --warmup 1 --no-cpu-baseline --no-hip-graph > $ROOT/$OUT/pmc_sq2_train_$name.log 2>&1); echo "pmc pass $name rc=$?"
                 (cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $ROOT/$OUT/pmc_sq2_generic/$name -o p -- python $ROOT/scripts/generic_rates.py --only surface_128x4_lvis_fp32,surface_128x4_lvis > $ROOT/$OUT/pmc_sq2_generic_$name.log 2>&1); echo "pmc pass $name rc=$?"
               done
               for w in train generic; do python scripts/pmc_digest.py $OUT/pmc_sq2_$w > $OUT/pmc_sq2_${w}_digest.json; python - <<PYEOF
import json
d=json.load(open("$OUT/pmc_sq2_${w}_digest.json"))
for k,v in d.items():
    if 'fused_kernel<1' in k or 'generic_bwd' in k or 'generic_wgrad_kernel' in k or 'generic_kernel' in k:
        wc=v.get('SQ_WAVE_CYCLES',1)
        print(k[:64], {a: round(b/wc,4) for a,b in v.items() if a.startswith('SQ_') and a!='SQ_WAVE_CYCLES'}, int(wc))
PYEOF
               done ;;
    fused-ab) for lib in ${AB_LIBS:-libnfx libnfx_xp6 libnfx_xp7}; do NFX_LIB_PATH=$ROOT/nerfactor_amd/$lib.so timeout 300 python bench.py --legs train --train-models nerfactor_microfacet --steps 20 --warmup 3 --no-cpu-baseline > $OUT/bench_train_$lib.json 2> $OUT/bench_train_$lib.err; python - <<PYEOF
import json
d=json.loads(open("$OUT/bench_train_$lib.json").read().strip().splitlines()[-1])
t=d.get('train',d)
print("$lib", json.dumps({k:v for k,v in (t.get('nerfactor_microfacet') or t).items() if not isinstance(v,(dict,list))})[:600])
PYEOF
               done ;;
    splits-ab) for sp in 64 128 256; do timeout 300 python scripts/generic_rates.py --only ${GENERIC_ONLY:-surface_128x4_lvis_fp32,nerf_enc_256x8_fp32,surface_128x4_lvis_fp32_native,surface_128x4_lvis} --option wgrad_splits=$sp > $OUT/generic_rates_splits$sp.json 2> $OUT/generic_rates_splits$sp.err; echo "splits $sp"; cat $OUT/generic_rates_splits$sp.json; tail -2 $OUT/generic_rates_splits$sp.err; done ;;
    ring-ab) for lib in ${RING_LIBS:-libnfx libnfx_xpA libnfx_xpB}; do NFX_LIB_PATH=$ROOT/nerfactor_amd/$lib.so timeout 300 python scripts/generic_rates.py > $OUT/generic_rates_$lib.json 2> $OUT/generic_rates_$lib.err; echo $lib; python - <<PYEOF
import json
d=json.load(open("$OUT/generic_rates_$lib.json"))
for k,v in d.items(): print('  %-34s fwd %7.3f ms %6.1f TF   bwd %7.3f ms %6.1f TF' % (k, v['fwd_ms'], v['fwd_tflops'], v['bwd_ms'], v['bwd_tflops']))
PYEOF
               tail -1 $OUT/generic_rates_$lib.err; done ;;
    step-trace) (cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d $ROOT/$OUT/step_trace -o t -- python $ROOT/bench.py --legs train --train-models ${TRACE_MODEL:-nerfactor_microfacet} --steps 3 --warmup 1 --no-cpu-baseline --no-hip-graph > $ROOT/$OUT/step_trace_run.log 2>&1); python scripts/step_trace.py $(find $OUT/step_trace -name "*kernel_trace.csv" | head -1) ${TRACE_MIN_US:-1000} > $OUT/step_trace_${TRACE_MODEL:-nerfactor_microfacet}.txt; tail -${TRACE_TAIL:-45} $OUT/step_trace_${TRACE_MODEL:-nerfactor_microfacet}.txt ;;
    *) echo "unknown stage $st" ;;
  esac
done
echo done $(date +%T)
