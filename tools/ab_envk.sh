#!/bin/bash
# kbench cases for several values of an environment knob: tools/ab_envk.sh VAR "v1 v2 ..." cases...
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; VALS=$2; shift 2
for val in $VALS; do
  export $V=$val
  echo "== $V=$val"
  timeout -k 10 200 python3 -u $R/tools/kbench.py "$@" --iters 50 2>/dev/null | cut -c1-40 || exit 1
done
