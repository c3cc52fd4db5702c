#!/bin/bash
# Winograd kernel ablations (GMK_WINO_DBG bits: 1 no epilogue, 4 first-chunk transforms only, 8 no MFMA section)
R=${GRAFT_REPO_ROOT:-/root/repo}
export GMK_WINO=3
for d in ${@:-0 1 8 13}; do
  export GMK_WINO_DBG=$d
  echo "== GMK_WINO_DBG=$d"
  timeout -k 10 100 python3 -u $R/tools/kbench.py fwd_s1_16 dgrad_s1_16 --iters 50 2>/dev/null | cut -c1-62 || exit 1
done
