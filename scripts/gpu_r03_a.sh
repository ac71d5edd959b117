#!/bin/bash
# Round 3, call A: (1) where the hipGraph step diverges from the eager one, (2) scripts/gpu_r03_first.sh (variants 9/10, lvis 9,
# loader-fed capture, fp32 labels).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03a
mkdir -p $OUT
rocm-smi --showclocks > $OUT/gpu.txt 2>&1
for cfg in "nerfactor_microfacet 0.01 --same-batch" "nerfactor_microfacet 0 --same-batch" "nerfactor_microfacet 0.01" "nerfactor 0.01 --same-batch" "shape 0.01 --same-batch"; do
  set -- $cfg
  timeout 150 python scripts/diag_graph_diverge.py --model $1 --jitter $2 ${3:-} --steps 120 > "$OUT/diverge_$1_$2${3:-}.json" 2> "$OUT/diverge_$1_$2${3:-}.err"
  echo "diverge $cfg rc=$?"; cut -c1-1500 "$OUT/diverge_$1_$2${3:-}.json"
done
TAG=r03a bash scripts/gpu_r03_first.sh
