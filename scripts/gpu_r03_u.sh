#!/bin/bash
# Round 3, call U: the learned-BRDF kernel with two waves per SIMD (default) — soak against the 4-wave form, the tests
# that touch it, the NeRFactor bench legs
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/r03u
mkdir -p $OUT
REPS=400 timeout 300 python scripts/brdf_nw8_soak.py 2>&1 | tail -3 | tee $OUT/soak.log
timeout 900 python -m pytest tests/test_gpu_nerfactor.py tests/test_gpu_reference_golden.py tests/test_gpu_train.py -q -m gpu -k "brdf or nerfactor or plugin or learned or variants" 2>&1 | tail -3 | tee $OUT/pytest_tail.txt
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs nerfactor_microfacet,nerfactor > $OUT/bench_nerfactor_legs.json 2> $OUT/bench.err
python - <<'PY'
import json
j = json.loads(open('gpurun_out/r03u/bench_nerfactor_legs.json').read().strip().splitlines()[-1])
for m, v in j['nerfactor'].items():
    print(m, 'ms/view %.2f' % v['ms_per_step'], 'rgb max-abs %.2e' % v['parity']['max_abs'], v.get('brdf_spec'))
PY
