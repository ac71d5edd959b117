#!/bin/bash
# Round 3, call C: (1) the graphed-step divergence the 200-step test found (nerfactor_microfacet, 256 rays, jitter on) — alone in a
# fresh process and step by step; (2) cycle stamps of the default NeRF kernel with parts of the tile switched off
# (timing + ablation build): what makes the first tile of every layer cost two tile times?
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03c
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_train.py -x -q -k "graphed_train_step and nerfactor_microfacet-0.01" > $OUT/pytest_graph_alone.log 2>&1; echo "alone rc=$?"; grep -h "first differing\|passed\|failed" $OUT/pytest_graph_alone.log | cut -c1-400
for cfg in "--rays 256 --batches 6 --steps 200" "--rays 256 --batches 6 --steps 200 --keep-vis --ids" "--rays 1024 --batches 8 --steps 200" "--rays 256 --same-batch --steps 200"; do
  tag=$(echo $cfg | tr -d ' -')
  timeout 200 python scripts/diag_graph_diverge.py --model nerfactor_microfacet $cfg > $OUT/diverge_$tag.json 2> $OUT/diverge_$tag.err
  python - <<PY
import json
try:
    j = json.load(open("$OUT/diverge_$tag.json"))
    print("$cfg", {k: j.get(k) for k in ("first_param_diff_step", "first_grad_diff_step", "first_nonfinite_graph", "loss_graph_differs_at")}, j.get("grad_diff_at_first", [])[:4], j["loss_eager_tail"], j["loss_graph_tail"])
except Exception as e:
    print("$cfg", "FAILED", e)
PY
done
if [ -f nerfactor_amd/libnfx_t.so ]; then
  for ab in 0 8 64 1 2 75; do
    NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_t.so NFX_ABLATE=$ab timeout 120 python scripts/v6_timing.py > $OUT/stamps_ab$ab.log 2>&1
    echo "ablate $ab: $(tail -2 $OUT/stamps_ab$ab.log | tr '\n' ' ')"
  done
fi
