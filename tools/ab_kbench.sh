#!/bin/bash
# same-call A/B of tools/kbench.py cases between library variants: tools/ab_kbench.sh <libA.so> <libB.so> <case-substrings...>
R=${GRAFT_REPO_ROOT:-/root/repo}
A=$1; B=$2; shift 2
for lib in $A $B $A $B; do
  export GMK_LIB_PATH=$R/$lib
  echo "== $lib"
  timeout -k 10 200 python3 -u $R/tools/kbench.py "$@" --iters 50 2>/dev/null | cut -c1-60 || exit 1
done
