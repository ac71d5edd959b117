#!/bin/bash
# First GPU call of round 3 (≈3 min): what round 2 wrote after its GPU budget was spent, checked on hardware.
#   1. the hipGraph capture fed from the dataset loaders (optim._detached: r02 crashed in capture_end)
#      — each configuration in its own process under `timeout`, so a crash costs one line, not the call;
#   2. the loader benchmark, all four configurations;
#   3. the labels of `bench.py --precision fp32` (a short run).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${TAG:-r03first}
mkdir -p $OUT
for cfg in hipGraph:0 hipGraph:2; do
  timeout 60 python -X faulthandler scripts/bench_loader.py --imh 128 --views 12 --epochs 2 --only $cfg \
      > $OUT/loader_$cfg.json 2> $OUT/loader_$cfg.err
  echo "$cfg rc=$? $(cut -c1-300 $OUT/loader_$cfg.json)"; grep -c "AccumulateGrad" $OUT/loader_$cfg.err
done
timeout 120 python scripts/bench_loader.py > $OUT/bench_loader.json 2> $OUT/bench_loader.err; echo "loader rc=$?"; cut -c1-900 $OUT/bench_loader.json
timeout 200 python bench.py --steps 3 --warmup 1 --precision fp32 --cpu-budget 6 > $OUT/bench_fp32.json 2> $OUT/bench_fp32.err
echo "bench fp32 rc=$?"; python - <<PY
import json
j = json.load(open("$OUT/bench_fp32.json"))
print(j["dtype"], j["value"], j["roofline"]["kernel"])
for k, v in j["nerfactor"].items():
    print(k, v["ms_per_step"], v["roofline"]["kernel"])
PY
