#!/bin/bash
# r02: learned-BRDF kernel with front-lit compaction — parity tests, then bench A/B of the variants on one box.
set -u
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/r02b
mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_nerfactor.py -m gpu -q -x 2>&1 | tail -15 | tee $OUT/pytest.log
for cfg in "3 4" "5 4" "5 3" "6 4" "6 3"; do
  set -- $cfg
  echo "=== NFX_BRDF_VARIANT=$1 NFX_BRDF_CT=$2" | tee -a $OUT/ab.log
  NFX_BRDF_VARIANT=$1 NFX_BRDF_CT=$2 timeout 300 python bench.py --steps 5 --warmup 2 --legs nerf,nerfactor --no-cpu-baseline 2>/dev/null \
    | python -c "import json,sys; d=json.loads(sys.stdin.readline()); n=d['nerfactor']['nerfactor']; print(json.dumps({'ms_per_step': n['ms_per_step'], 'brdf_spec': n['brdf_spec'], 'lvis_ms': n['roofline']['avg_launch_ms']}))" | tee -a $OUT/ab.log
done
