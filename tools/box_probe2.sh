#!/bin/bash
# bench.py's own step time against tools/step_stats.py windows, thin-channel vector-ALU kernels off / on, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showserial 2>/dev/null | grep -i "serial number:"
for rep in 1 2; do
for v in "0 0" "3 1"; do
  set -- $v; export GMK_THIN_IN=$1 GMK_THIN_OUTV=$2
  echo -n "thin $1 $2 bench.py: "; timeout -k 10 200 python3 $R/bench.py --no-cpu --no-fid 2>/dev/null | python3 -c "import sys,json; print(json.loads(sys.stdin.read())['ms_per_step'])" || exit 1
  echo -n "thin $1 $2 step_stats: "; timeout -k 10 120 python3 $R/tools/step_stats.py 3 100 2>/dev/null | tail -1 || exit 1
done
done
