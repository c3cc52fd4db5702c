#!/bin/bash
# step windows for thin-channel kernels off / on x 1..3 weight-gradient streams, same box
R=${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showserial 2>/dev/null | grep -i "serial number:"
for v in "0 0" "3 1"; do
  set -- $v; export GMK_THIN_IN=$1 GMK_THIN_OUTV=$2
  for ss in 1 2 3; do
    export GMK_SIDE_STREAMS=$ss
    echo -n "thin $1 $2 side $ss: "; timeout -k 10 120 python3 $R/tools/step_stats.py 3 100 2>/dev/null | tail -1 || exit 1
  done
done
