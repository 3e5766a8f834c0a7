#!/bin/bash
# one gpurun call: flow-kernel parity tests + per-rank and small-mesh timings, stage launches vs the dataflow launch
mkdir -p gpurun_out/r03a
O=gpurun_out/r03a
timeout 900 python -m pytest tests/test_gpu_flow_kernel.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -5 $O/tests.log
for m in 2 4 8; do
  for fl in 0 1; do
    timeout 300 python tools/rankbench.py --world 8 --rank 3 --every $m --exchange p2p --nosplit --flow $fl --steps 240 2>&1 | tail -1 >> $O/rank.log
  done
done
for fl in 0 1; do
  timeout 300 python tools/rankbench.py --world 16 --rank 7 --every 4 --exchange p2p --nosplit --flow $fl --steps 240 2>&1 | tail -1 >> $O/rank.log
done
cat $O/rank.log
for nx in 125 177 250 300; do
  for fl in 0 1; do
    THETIS_AMD_FLOW=$fl THETIS_AMD_FUSED_STEP=0 timeout 300 python tools/kbench.py --nx $nx --ny $((nx/2)) --steps 96 --tag flow$fl 2>&1 | tail -1 >> $O/kbench.log
  done
done
cat $O/kbench.log
