#!/bin/bash
# Round 3, call B: the whole GPU suite on the new defaults (fp32-class normal head, 200-step graphed-step test, 800^2 NeRFactor
# frame test), smoke(), the r02 hipGraph benchmark command that ended in NaN, the step-by-step diagnostic with that
# benchmark's seeds, and the new bench line (train + OLAT legs).
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/r03b
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -x -q -s > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $OUT/pytest_gpu.log
grep -h "800x800 frame" $OUT/pytest_gpu.log
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; grep smoke: $OUT/smoke.log
for m in nerfactor_microfacet nerfactor; do
  timeout 120 python scripts/bench_train.py --model $m --graph --steps 100 > $OUT/train_graph_$m.json 2> $OUT/train_graph_$m.err; echo "bench_train --graph $m rc=$?"; cat $OUT/train_graph_$m.json; tail -2 $OUT/train_graph_$m.err
  timeout 120 python scripts/bench_train.py --model $m --steps 103 > $OUT/train_eager_$m.json 2> $OUT/train_eager_$m.err; echo "bench_train eager $m rc=$?"; cat $OUT/train_eager_$m.json; tail -2 $OUT/train_eager_$m.err
  timeout 150 python scripts/diag_graph_diverge.py --model $m --seed 5 --data-seed 100 --same-batch --steps 140 > $OUT/diverge_bench_$m.json 2> $OUT/diverge_bench_$m.err
  python - <<PY
import json
j = json.load(open("$OUT/diverge_bench_$m.json"))
print("$m", {k: j[k] for k in ("first_param_diff_step", "first_nonfinite_graph", "first_nonfinite_eager", "loss_graph_differs_at")}, j["loss_eager"][95:115])
PY
done
timeout 600 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"; tail -3 $OUT/bench.err
python - <<PY
import json
j = json.load(open("$OUT/bench.json"))
print("value", j["value"], "ms", j["ms_per_step"], "roofline", j["roofline"]["achieved"], j["roofline"]["frac"])
print("parity", j.get("parity")); print("fitted", j.get("parity_fitted_weights"))
for k, v in j.get("nerfactor", {}).items():
    print(k, v["ms_per_step"], v["roofline"]["achieved"], v.get("parity"))
for k, v in j.get("train", {}).items():
    print("train", k, v["ms_per_step"], v["roofline"], v["first_loss"], v["final_loss"])
print("olat", j.get("olat"))
PY
