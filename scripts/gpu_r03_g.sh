#!/bin/bash
# Round 3, call G: the operand fence (mlp_engine.hpp) in every kernel — A/B against the same sources built with
# -DNFX_NO_OPERAND_FENCE (alternating runs on one box), bit-identity of the two builds, then the whole GPU suite.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03g
mkdir -p $OUT
NF=$PWD/nerfactor_amd/libnfx_nofence.so
for rep in 1 2; do
  for lib in fence nofence; do
    if [ $lib = nofence ]; then export NFX_LIB_PATH=$NF; else unset NFX_LIB_PATH; fi
    timeout 200 python bench.py --legs nerf,nerfactor_microfacet,nerfactor --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err
    python - <<PY
import json
j = json.load(open("$OUT/bench_${lib}_$rep.json"))
n = j["nerfactor"]
print("$lib $rep: nerf %.3f M rays/s, mlp %.1f TF | lvis %.2f ms (%.0f TF) render mf %.2f ms | learned render %.2f ms, brdf_spec %.2f ms" % (
    j["value"] / 1e6, j["roofline"]["achieved"], n["nerfactor_microfacet"]["roofline"]["avg_launch_ms"], n["nerfactor_microfacet"]["roofline"]["achieved"],
    n["nerfactor_microfacet"]["ms_per_step"], n["nerfactor"]["ms_per_step"], n["nerfactor"]["brdf_spec"]["avg_launch_ms"]))
PY
  done
done
for lib in fence nofence; do
  if [ $lib = nofence ]; then export NFX_LIB_PATH=$NF; else unset NFX_LIB_PATH; fi
  for m in nerfactor_microfacet nerf; do
    timeout 120 python scripts/bench_train.py --model $m --steps 60 2>/dev/null | python -c "import sys, json; j = json.loads(sys.stdin.read()); print('$lib train $m %.3f ms/step loss %.6f' % (j['ms_per_step'], j['final_loss']))"
  done
  timeout 100 python scripts/bench_geometry.py 2>/dev/null | cut -c1-300
done
unset NFX_LIB_PATH
timeout 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
