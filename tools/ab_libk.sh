#!/bin/bash
# kbench cases for several libgmk builds: tools/ab_libk.sh "libA.so libB.so ..." cases...
R=${GRAFT_REPO_ROOT:-/root/repo}
LIBS=$1; shift
for lib in $LIBS $LIBS; do
  export GMK_LIB_PATH=$R/$lib
  echo "== $lib"
  timeout -k 10 200 python3 -u $R/tools/kbench.py "$@" --iters 50 2>/dev/null | cut -c1-40 || exit 1
done
