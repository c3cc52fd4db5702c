#!/bin/bash
# One call producing the evidence bench.py's JSON line is checked against (run on the GPU box from the repo root):
#   tools/prof_all.sh <tag>   ->  gpurun_out/<tag>/{kt_kernel_stats.csv, kt_kernel_trace.csv, step_share.json, roofline_legs.csv, pmc/summary.json}
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1
bash $R/tools/prof_step.sh $TAG 20 > /dev/null 2>&1 || { echo "prof_step failed"; exit 1; }
TR=$(find $R/gpurun_out/$TAG -name "*kernel_trace.csv" | head -1)
python3 $R/tools/step_share.py $TR $R/gpurun_out/$TAG/step_share.json > $R/gpurun_out/$TAG/step_share.txt 2>&1 || exit 1
python3 $R/tools/roofline_from_trace.py $TR > $R/gpurun_out/$TAG/roofline_legs.csv 2>&1 || exit 1
python3 $R/tools/step_timeline.py $TR 3 > $R/gpurun_out/$TAG/timeline.txt 2>&1
tail -c 1500 $R/gpurun_out/$TAG/bench.log
cat $R/gpurun_out/$TAG/roofline_legs.csv
bash $R/tools/pmc_targets.sh $TAG/pmc > $R/gpurun_out/$TAG/pmc.log 2>&1 || { echo "pmc failed"; tail -5 $R/gpurun_out/$TAG/pmc.log; exit 1; }
tail -3 $R/gpurun_out/$TAG/pmc.log | cut -c1-300
