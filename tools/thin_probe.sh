#!/bin/bash
# per-kernel durations (rocprofv3 kernel stats of bench.py's step) and step windows with the vector-ALU thin-channel kernels off / on
R=${GRAFT_REPO_ROOT:-/root/repo}
rocm-smi --showserial 2>/dev/null | grep -i "serial number:"
export GMK_THIN_IN=0 GMK_THIN_OUTV=0; bash $R/tools/prof_step.sh thinA 40 > /dev/null 2>&1
export GMK_THIN_IN=3 GMK_THIN_OUTV=1; bash $R/tools/prof_step.sh thinB 40 > /dev/null 2>&1
python3 $R/tools/kcmp.py $R/gpurun_out/thinA/kt_kernel_stats.csv $R/gpurun_out/thinB/kt_kernel_stats.csv
export GMK_THIN_IN=0 GMK_THIN_OUTV=0; python3 $R/tools/step_stats.py 3 100 | tail -1
export GMK_THIN_IN=3 GMK_THIN_OUTV=1; python3 $R/tools/step_stats.py 3 100 | tail -1
