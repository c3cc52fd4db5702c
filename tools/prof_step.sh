#!/bin/bash
# rocprofv3 kernel trace + stats of bench.py's train step (program directly after `--`): tools/prof_step.sh <name> [steps]
# Run on the GPU box from the repo root; summary lands in gpurun_out/<name>/kt_kernel_stats.csv
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# the profiler's host overhead would decide bench.py's per-device path tuning: profile the default paths unless told otherwise
export GMK_TUNE=${GMK_TUNE:-0}
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $OUT -o kt --output-format csv -- python3 $R/bench.py --steps ${2:-20} --warmup 5 --no-cpu --no-fid --no-secondary > $OUT/bench.log 2>&1
tail -c 700 $OUT/bench.log
