#!/bin/bash
# kbench cases for several settings of TWO environment knobs: tools/ab_env2k.sh VAR1 VAR2 "a1,b1 a2,b2 ..." cases...
R=${GRAFT_REPO_ROOT:-/root/repo}
V1=$1; V2=$2; VALS=$3; shift 3
for val in $VALS; do
  export $V1=${val%,*} $V2=${val#*,}
  echo "== $V1=${val%,*} $V2=${val#*,}"
  timeout -k 10 200 python3 -u $R/tools/kbench.py "$@" --iters 50 2>/dev/null | cut -c1-40 || exit 1
done
