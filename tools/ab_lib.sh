#!/bin/bash
# same-call A/B of two libgmk builds: kbench cases + windowed step statistics, A B A B.  tools/ab_lib.sh <libA.so> <libB.so> [kbench cases...]
R=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; shift 2
for lib in $A $B $A $B; do
  export GMK_LIB_PATH=$R/$lib
  echo "== $lib"
  if [ $# -gt 0 ]; then timeout -k 10 200 python3 -u $R/tools/kbench.py "$@" --iters 50 2>/dev/null | cut -c1-40 || exit 1; fi
  timeout -k 10 200 python3 -u $R/tools/step_stats.py 6 100 2>/dev/null | tail -1 || exit 1
done
