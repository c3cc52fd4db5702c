#!/bin/bash
# same-call A/B of an environment knob: tools/ab_env.sh VAR A B [rounds] -- bench.py step, A B A B ...
R=${GRAFT_REPO_ROOT:-/root/repo}
V=$1; A=$2; B=$3
for i in $(seq 1 ${4:-3}); do
  for val in $A $B; do
    export $V=$val
    echo -n "$V=$val: "
    timeout -k 10 200 python3 -u $R/bench.py --no-cpu --no-fid --steps 300 --warmup 50 2>/dev/null | tail -1 | python3 -c "import sys,json; print('step ms', json.loads(sys.stdin.read())['ms_per_step'])" || exit 1
  done
done
