#!/bin/bash
# kernel-trace of the FC GEMMs for a list of library variants: tools/fc_abl.sh <outdir> <lib.so> ...
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  n=$(basename $lib .so)
  export GMK_LIB_PATH=$R/$lib
  timeout -s KILL 120 rocprofv3 --kernel-trace -d $OUT/$n -o kt --output-format csv -- python3 $R/tools/kbench.py fc_ > $OUT/$n.log 2>&1 || { echo "$n failed"; tail -5 $OUT/$n.log; exit 1; }
done
python3 - <<PY
import csv, collections, glob, os
for d in sorted(glob.glob("$OUT/*/kt_kernel_trace.csv")):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(d)):
        if "gemm" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:44], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(os.path.basename(os.path.dirname(d)))
    for k, v in agg.items():
        print("  ", k, len(v), round(sum(v) / len(v), 2), min(v))
PY
