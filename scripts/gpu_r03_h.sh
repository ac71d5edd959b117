#!/bin/bash
# Round 3, call H: operand fence as a NON-volatile asm (the compiler may schedule it) against the no-fence build.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03h
mkdir -p $OUT
NF=$PWD/nerfactor_amd/libnfx_nofence.so
for rep in 1 2; do
  for lib in fence nofence; do
    if [ $lib = nofence ]; then export NFX_LIB_PATH=$NF; else unset NFX_LIB_PATH; fi
    timeout 200 python bench.py --legs nerf,nerfactor_microfacet --steps 4 --warmup 2 --no-cpu-baseline > $OUT/bench_${lib}_$rep.json 2> $OUT/bench_${lib}_$rep.err
    python - <<PY
import json
j = json.load(open("$OUT/bench_${lib}_$rep.json"))
n = j["nerfactor"]
print("$lib $rep: nerf %.3f M rays/s, mlp %.1f TF | lvis %.2f ms (%.0f TF) render mf %.2f ms" % (
    j["value"] / 1e6, j["roofline"]["achieved"], n["nerfactor_microfacet"]["roofline"]["avg_launch_ms"], n["nerfactor_microfacet"]["roofline"]["achieved"],
    n["nerfactor_microfacet"]["ms_per_step"]))
PY
  done
done
