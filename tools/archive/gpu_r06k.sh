#!/bin/bash
set -u
TAG=r06k
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
i=0
for order in 1 2k1 2c1 h2cs4k2k1 p2ks8ks4k1 1k1; do
  i=$((i+1))
  echo "=== order $order" >> $O/dbg.txt
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port $((29655+i)) tools/archive/dbg_fx1.py $order 2>&1 | grep -E "^rank|Error|error|Traceback" >> $O/dbg.txt
done
cat $O/dbg.txt | cut -c1-330
