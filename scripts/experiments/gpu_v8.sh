#!/bin/bash
# variant 5 vs 6 vs 8 on one box (default build and the build with variant 6/8 in MFMA VGPR form)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
python -m pytest tests/test_gpu_nerf.py -m gpu -q -k "bit_identical or batch_independence or (vs_oracle and (6 or 8))" 2>&1 | tail -3
for v in 5 6 8 5; do
  echo "== default build, variant $v"
  NFX_NERF_VARIANT=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['fine_launch_tflops'])"
done
for v in 6 8; do
  echo "== vgpr-form build, variant $v"
  NFX_LIB_PATH=$PWD/nerfactor_amd/libnfx_vf.so NFX_NERF_VARIANT=$v python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'], d['roofline']['fine_launch_tflops'])"
done
