#!/bin/bash
# tools/ab_env3.sh VAR v1 v2 v3 ... : tools/step_stats.py (6 windows of 100 steps) for each value, the list repeated 3 times
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
for i in 1 2 3; do
  for val in "$@"; do
    export $V=$val
    echo -n "$V=$val: "
    timeout -k 10 200 python3 -u $R/tools/step_stats.py 6 100 2>/dev/null | tail -1 || exit 1
  done
done
