#!/bin/bash
O=gpurun_out/r03e; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_flow_kernel.py -x -q -m gpu > $O/tests.log 2>&1; echo "tests rc=$?" >> $O/tests.log
tail -3 $O/tests.log
for nx in 125 354; do
  THETIS_AMD_FLOW=1 THETIS_AMD_FUSED_STEP=0 timeout 300 python tools/kbench.py --nx $nx --ny $((nx/2)) --steps 96 --tag flow2 2>&1 | tail -1 >> $O/ab.log
done
for m in 1 2; do
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every $m --exchange p2p --nosplit --flow 1 --steps 240 2>&1 | tail -1 >> $O/ab.log
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every $m --exchange p2p --nosplit --flow 1 --graph-mode full --steps 240 2>&1 | tail -1 >> $O/ab.log
done
timeout 300 python tools/rankbench.py --world 8 --rank 3 --every 2 --exchange stub --flow 1 --steps 240 2>&1 | tail -1 >> $O/ab.log
cat $O/ab.log
