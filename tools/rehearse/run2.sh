#!/bin/bash
# 2-rank rehearsal of bench.py on ONE GPU over gloo (transport differs from RCCL, everything above it is the product path)
R=${GRAFT_REPO_ROOT:-/root/repo}
export GMK_DIST_BACKEND=gloo GMK_FORCE_DEVICE=0 PYTHONUNBUFFERED=1 PYTHONPATH=$R/tools/rehearse
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
timeout -k 10 ${TMO:-90} python3 -u -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 \
  --log-dir $OUT --redirects 3 $R/bench.py --gpus 2 "$@" > $OUT/out.txt 2>&1
echo "rc=$?"
for f in $(find $OUT -name "stdout.log" | sort); do echo "== $f"; tail -c 600 $f; echo; done
for f in $(find $OUT -name "stderr.log" | sort); do echo "== $f"; grep -A8 "Timeout\|Error\|error" $f | head -24; done
