#!/bin/bash
# same-call A/B of bench.py's step between library variants: tools/ab_step.sh <libA.so> <libB.so> [rounds]
# (devices of the pool and a device over time differ by a few %: only interleaved runs inside one call compare)
R=${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 ${3:-2}); do
  for lib in $1 $2; do
    export GMK_LIB_PATH=$R/$lib
    echo -n "$lib: "
    timeout -k 10 200 python3 -u $R/bench.py --no-cpu --no-fid --steps 300 --warmup 30 2>/dev/null | tail -1 | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" || exit 1
  done
done
