#!/bin/bash
# rocprofv3 counter passes over tools/${KB_SCRIPT:-kbench.py} cases (program directly after `--`): tools/pmc_kbench.sh <outdir> <case> [<case> ...]
set -u
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout -s KILL 120 rocprofv3 --kernel-trace --pmc $grp -d $OUT/p$i -o p$i --output-format csv -- python3 $R/tools/${KB_SCRIPT:-kbench.py} "$@" --iters 5 > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; exit 1; }
done
python3 $R/tools/pmc_summary.py $OUT/summary.json $(find $OUT -name "*counter_collection.csv" | sort) > $OUT/summary.txt 2>&1
grep -v "at::native\|distribution" $OUT/summary.txt | cut -c1-1200
