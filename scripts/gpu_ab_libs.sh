#!/bin/bash
# bench.py with the default library and several experiment builds (ALTS="a.so b.so"), two rounds, same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for i in 1 2; do
  for lib in default $ALTS; do
    if [ $lib = default ]; then unset NFX_LIB_PATH; else export NFX_LIB_PATH=$PWD/$lib; fi
    echo -n "$lib: "; python bench.py --steps 4 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['achieved'])"
  done
done
