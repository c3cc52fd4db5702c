#!/bin/bash
# What the DP machinery costs one rank before any xGMI traffic: bench.py's timed step with and without a one-rank RCCL
# group under it (GMK_DP_REHEARSE=1: three async all-reduces per step on RCCL's stream, the launch plan replayed in
# segments), same box, same call, alternating.
#   gpurun -- 'bash tools/rehearse/rccl_world1.sh > gpurun_out/rccl1.txt 2>&1'
set -e
cd "$(dirname "$0")/../.."
for rep in 1 2; do
  for mode in 0 1; do
    echo "== GMK_DP_REHEARSE=$mode rep $rep"
    GMK_DP_REHEARSE=$mode timeout -k 10 170 python bench.py --steps 300 --warmup 30 --no-cpu --no-fid --no-secondary \
      | python -c 'import sys,json; o=json.loads(sys.stdin.readlines()[-1]); print(o["ms_per_step"], o["value"], o.get("dp_rehearsal","plain"))'
  done
done
