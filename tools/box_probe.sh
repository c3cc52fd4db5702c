#!/bin/bash
# which device / host is this, how long does the host take to enqueue a step, and the step with the vector-ALU thin-channel kernels off / on
R=${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showserial 2>/dev/null | grep -i "serial number:"
lscpu | grep -E "Model name|^CPU MHz|BogoMIPS" | head -3
timeout -k 10 120 python3 $R/tools/hosttime.py 2>/dev/null | head -3
export GMK_THIN_IN=0 GMK_THIN_OUTV=0; echo -n "thin off: "; timeout -k 10 120 python3 $R/tools/step_stats.py 3 100 | tail -1
export GMK_THIN_IN=3 GMK_THIN_OUTV=1; echo -n "thin on:  "; timeout -k 10 120 python3 $R/tools/step_stats.py 3 100 | tail -1
