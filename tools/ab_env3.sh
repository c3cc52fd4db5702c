#!/bin/bash
# tools/ab_env3.sh VAR v1 v2 v3 ... : bench.py step for each value, the whole list repeated 3 times (first pass = warm-up of the device)
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; shift
for i in 1 2 3; do
  for val in "$@"; do
    export $V=$val
    echo -n "$V=$val: "
    timeout -k 10 200 python3 -u $R/bench.py --no-cpu --no-fid --steps 300 --warmup 50 2>/dev/null | tail -1 | python3 -c "import sys,json; print('step ms', json.loads(sys.stdin.read())['ms_per_step'])" || exit 1
  done
done
