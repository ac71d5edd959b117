#!/bin/bash
# scripts/bench_nerfactor.py with the default library and experiment builds (ALTS), two rounds, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  for lib in default $ALTS; do
    if [ $lib = default ]; then unset NFX_LIB_PATH; else export NFX_LIB_PATH=$PWD/$lib; fi
    for m in nerfactor_microfacet nerfactor; do
      echo -n "$lib $m: "; python scripts/bench_nerfactor.py --model $m --steps 5 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v, 3) for k, v in d.items() if isinstance(v, float)})" | cut -c1-300
    done
  done
done
