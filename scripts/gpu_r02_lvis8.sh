#!/bin/bash
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r02e
mkdir -p $OUT
timeout 300 python -m pytest tests/test_gpu_nerfactor.py -m gpu -q -x -k "variants or lvis" 2>&1 | tail -5 | tee $OUT/pytest.log
for v in 4 8 4 8; do
  echo "=== NFX_LVIS_VARIANT=$v" | tee -a $OUT/ab.log
  NFX_LVIS_VARIANT=$v timeout 300 python bench.py --steps 5 --warmup 2 --legs nerfactor_microfacet --no-cpu-baseline 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); n=d['nerfactor']['nerfactor_microfacet']; print(json.dumps({'ms_per_step': n['ms_per_step'], 'lvis_ms': n['roofline']['avg_launch_ms'], 'tflops': n['roofline']['achieved']}))" | tee -a $OUT/ab.log
done
